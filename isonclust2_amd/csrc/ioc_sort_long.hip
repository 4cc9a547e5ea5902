// ioc_sort_long.hip — distinct values of queries beyond the in-LDS sort (iock_distinct_long).  A file of its own: rocPRIM's segmented
// sort is 6 MB of code object, and a code object is loaded when one of its kernels is first launched — the MinDB export of every
// `cluster` process (ioc_sort.hip) must not pay for a path that only a batch with a 35 kb read takes.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>

#include "ioc_kernels.h"

namespace {
constexpr int BLK = 256;
}

// ---- queries beyond the in-LDS sort of k_distinct_radix (more than IOC_DISTINCT_LDS_MAX forward minimizers: reads of ~35 kb and
// up, chimeric ultra-long ones of hundreds of kb) --------------------------------------------------------------------------------
// Their values are gathered side by side, sorted segment by segment with rocPRIM (library code for a rare case: the reference's
// std::set has no limit, minimizer.cpp:31-42, so neither has this path), and written out like the short queries': distinct values,
// their count, and — sorted index build — the (value, target) pairs with the sentinel behind them.
namespace {

__global__ void __launch_bounds__(BLK)
k_long_gather(const int32_t* __restrict__ qid, const int64_t* __restrict__ off_fwd, const unsigned long long* __restrict__ seg, const uint32_t* __restrict__ mins,
              uint32_t* __restrict__ out)
{
    const int j = qid[blockIdx.x];
    const int64_t b = off_fwd[j];
    const uint32_t m = uint32_t(off_fwd[j + 1] - b);
    uint32_t* o = out + seg[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < m; i += BLK) o[i] = mins[b + i];
}

__global__ void __launch_bounds__(BLK)
k_long_unique(const int32_t* __restrict__ qid, const unsigned long long* __restrict__ seg, const uint32_t* __restrict__ sorted, const int64_t* __restrict__ doff,
              uint32_t* __restrict__ dvals, uint32_t* __restrict__ dcount, uint32_t* __restrict__ pk, void* __restrict__ pv, int pv16, uint32_t target0,
              uint32_t sentinel)
{
    __shared__ uint32_t wsum[BLK / 64];
    __shared__ uint32_t s_base;
    const int j = qid[blockIdx.x];
    const uint32_t* s = sorted + seg[blockIdx.x];
    const uint32_t m = uint32_t(seg[blockIdx.x + 1] - seg[blockIdx.x]);
    uint32_t* out = dvals + doff[j];
    uint32_t* pko = pk ? pk + doff[j] : nullptr;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t c = 0; c < m; c += BLK) {
        const uint32_t i = c + threadIdx.x;
        const bool flag = i < m && (i == 0 || s[i] != s[i - 1]);
        const unsigned long long bm = __ballot(flag);
        if (lane == 0) wsum[wave] = uint32_t(__popcll(bm));
        __syncthreads();
        uint32_t before = s_base;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (flag) {
            const uint32_t at = before + uint32_t(__popcll(bm & ((1ull << lane) - 1ull)));
            out[at] = s[i];
            if (pko) pko[at] = s[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < BLK / 64; ++w) t += wsum[w];
            s_base += t;
        }
        __syncthreads();
    }
    const uint32_t base = s_base;
    if (threadIdx.x == 0) dcount[j] = base;
    if (pko) {
        for (uint32_t d = base + threadIdx.x; d < m; d += BLK) pko[d] = sentinel;
        const uint32_t t = target0 + uint32_t(j);
        if (pv16) {
            uint16_t* o = static_cast<uint16_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += BLK) o[d] = uint16_t(t);
        } else {
            uint32_t* o = static_cast<uint32_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += BLK) o[d] = t;
        }
    }
}

}  // namespace

size_t iock_distinct_long_temp(size_t total, uint32_t nlong, int value_bits)
{
    size_t a = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, unsigned(total), nlong, (const unsigned long long*)nullptr,
                                             (const unsigned long long*)nullptr, 0u, unsigned(value_bits), hipStream_t(nullptr));
    return a + 256;
}

// work = [total words: gathered][total words: sorted][temp]; d_qid [nlong], d_seg [nlong + 1] (offsets of the segments in the gathered array)
hipError_t iock_distinct_long(hipStream_t st, uint32_t nlong, size_t total, const int32_t* d_qid, const unsigned long long* d_seg, const int64_t* off_fwd,
                              const uint32_t* mins, const int64_t* doff, uint32_t* dvals, uint32_t* dcount, int value_bits, uint32_t* work, void* temp,
                              size_t temp_bytes, uint32_t* pk, void* pv, int pv16, uint32_t target0, uint32_t sentinel)
{
    if (!nlong) return hipSuccess;
    uint32_t* in = work;
    uint32_t* out = work + total;
    hipLaunchKernelGGL(k_long_gather, dim3(nlong), dim3(BLK), 0, st, d_qid, off_fwd, d_seg, mins, in);
    size_t tb = temp_bytes;
    const int bits = value_bits < 1 || value_bits > 32 ? 32 : value_bits;
    hipError_t e = rocprim::segmented_radix_sort_keys(temp, tb, in, out, unsigned(total), nlong, d_seg, d_seg + 1, 0u, unsigned(bits), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_long_unique, dim3(nlong), dim3(BLK), 0, st, d_qid, d_seg, out, doff, dvals, dcount, pk, pv, pv16, target0, sentinel);
    return hipGetLastError();
}
