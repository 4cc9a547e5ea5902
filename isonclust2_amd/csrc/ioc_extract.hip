// placeholder — replaced below in this round by the sort-stage kernels
#include "ioc_internal.h"
extern "C" {
int ioc_qual_scores(ioc_ctx* c, int32_t, const int64_t*, const uint8_t*, int32_t, double*, double*) { return ioc_fail(c, IOC_ERR_STATE, "not built"); }
int ioc_extract_minimizers(ioc_ctx* c, int32_t, const int64_t*, const uint8_t*, const uint8_t*, int32_t, int32_t, uint32_t*, double*, int64_t*, int64_t*, int32_t*) { return ioc_fail(c, IOC_ERR_STATE, "not built"); }
int ioc_extracted_download(ioc_ctx* c, uint32_t*, uint32_t*, int64_t) { return ioc_fail(c, IOC_ERR_STATE, "not built"); }
int ioc_queries_from_extracted(ioc_ctx* c, const uint8_t*, const uint8_t*, const uint32_t*) { return ioc_fail(c, IOC_ERR_STATE, "not built"); }
}
