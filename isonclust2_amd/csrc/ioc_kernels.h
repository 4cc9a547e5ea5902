// ioc_kernels.h — launcher prototypes of ioc_kernels.hip / ioc_extract.hip (internal).
#ifndef IOC_KERNELS_H
#define IOC_KERNELS_H

#include <hip/hip_runtime_api.h>

#include <climits>
#include <cstddef>
#include <cstdint>

// Arguments of k_decide (getBestClusterMapping + getMappedRatio, src/cluster.cpp:324-406).
struct DecideArgs {
    int n;
    uint32_t L;
    int first;  // queries < first are final
    const int64_t* off_fwd;
    const int64_t* off_rev;
    const uint32_t* mins;
    const uint32_t* pos;
    const uint32_t* hpc_len;
    const uint8_t* err_cell;
    const uint32_t* min_total;
    const uint8_t* left_err;
    const int64_t* doff;
    const uint32_t* dvals;
    const uint32_t* dcount;
    const int64_t* lset_off;
    const uint32_t* lset_val;
    const uint32_t* cand_key;
    const uint32_t* cand_size;
    uint32_t* cand_mapped;
    const uint32_t* cand_count;
    const int32_t* glim;  // [15][15]
    const uint8_t* valid_in;
    uint8_t* valid_out;
    int32_t* dec_target;
    int8_t* dec_strand;
    uint8_t* flags;
    const int32_t* forced_t;  // INT32_MIN = not forced
    const int8_t* forced_s;
    uint32_t* first_changed;  // atomicMin
    unsigned long long* n_evals;
    int min_shared;
    double min_fraction;
    int32_t* cut;           // per query: cut of the candidate walk, INT32_MAX = no walk
    uint32_t* q_items;      // work queue of (query, candidate index) pairs awaiting k_eval
    uint32_t* q_count;
    uint32_t q_cap;
    uint32_t* incomplete;   // queries whose walk met a candidate that is not evaluated yet
    int phase;              // 1: candidates of maximal Size only, 2: the rest of the walk
    uint32_t* top;          // per query: top Size over the targets that are clusters
    uint8_t* done;          // per query: decided in phase 1
    unsigned long long* diag;  // diagnostic cycle sums (IOC_EVAL_DIAG), normally nullptr
    int lazy;               // lazy sweep: only the maximal-Size candidates are walked (see ioc_resolve)
    // upper bound of totalMapped (k_gap_bounds): [query][strand][error cell of the target] -> (D, head + tail); null = no bound test
    const uint2* gap_bound;
    int own_stride, own_offset;  // sharded merge: this rank decides the queries j with j % own_stride == own_offset (stride <= 1: all)
    // alignment fallback (sahlin / furious): verdict of the alignment for a query, used only if its
    // mapping walk finds nothing although top >= MinShared (INT32_MIN = none yet, -1 = no hit either);
    // the candidates tied at the top Size (the ones getBestClusterAln tries) are reported per query
    const int32_t* aln_t;
    const int8_t* aln_s;
    uint32_t* tie_count;    // per query: number of top-Size candidates that are clusters
    uint32_t* tie_keys;     // per query: up to IOC_TIE_SLOTS of their keys (target << 1 | strand bit), any order
    // the walk of a query as k_decide_scan found it (the candidates that are clusters and pass the Size rule of the phase), for
    // k_decide_pick of the same phase: walk_n[j] entries of walk_c[j * IOC_WALK_SLOTS ...], or IOC_WALK_OVERFLOW (pick scans the list)
    uint32_t* walk_n;
    uint32_t* walk_c;
};
#define IOC_WALK_SLOTS 32
#define IOC_WALK_OVERFLOW 0xFFFFFFFFu
#ifndef IOC_TIE_SLOTS
#define IOC_TIE_SLOTS 16  // (include/isonclust2_hip.h)
#endif

extern "C" {
// pk / pv (optional): the sorted index build's (value, target) pairs of the queries, target0 + j for query j, `sentinel` keys behind a
// query's distinct values; *pairs_written says whether the kernel that ran wrote them (the bitonic fallback does not)
hipError_t iock_distinct(hipStream_t st, int n, const int64_t* off_fwd, const uint32_t* mins, const int64_t* doff,
                         uint32_t* dvals, uint32_t* dcount, uint32_t pmax, int value_bits, uint32_t* pk = nullptr, void* pv = nullptr,
                         int pv16 = 0, uint32_t target0 = 0, uint32_t sentinel = 0, int* pairs_written = nullptr);
// one per kernel file: loads the file's code object (ioc_ctx_prewarm)
hipError_t iock_warm_kernels();
hipError_t iock_warm_build_sort();
hipError_t iock_warm_score();
hipError_t iock_warm_resolve();
hipError_t iock_warm_sort();
hipError_t iock_warm_align();
// queries of more than IOC_DISTINCT_LDS_MAX forward minimizers (k_distinct_radix skips them): ioc_sort_long.hip
#define IOC_DISTINCT_LDS_MAX 8192u
size_t iock_distinct_long_temp(size_t total, uint32_t nlong, int value_bits);
hipError_t iock_distinct_long(hipStream_t st, uint32_t nlong, size_t total, const int32_t* d_qid, const unsigned long long* d_seg, const int64_t* off_fwd,
                              const uint32_t* mins, const int64_t* doff, uint32_t* dvals, uint32_t* dcount, int value_bits, uint32_t* work, void* temp,
                              size_t temp_bytes, uint32_t* pk, void* pv, int pv16, uint32_t target0, uint32_t sentinel);
hipError_t iock_hash_insert_queries(hipStream_t st, int n, const int64_t* doff, const uint32_t* dvals,
                                    const uint32_t* dcount, uint32_t* keys, uint32_t cap, uint32_t shift,
                                    uint32_t* cnt, uint32_t* dslot, uint32_t* dpos, uint32_t* err);
hipError_t iock_hash_insert_left(hipStream_t st, int64_t nkeys, const uint32_t* lkeys, const int64_t* loffs,
                                 uint32_t* keys, uint32_t cap, uint32_t shift, uint32_t* cnt, uint32_t* lslot,
                                 uint32_t* err);
// up to IOC_FILL_SEGS buffers (16-byte aligned, whole 32-bit words) filled with a 32-bit value each by one launch
#define IOC_FILL_SEGS 6
hipError_t iock_fill_multi(hipStream_t st, int nseg, void* const* ptrs, const size_t* bytes, const uint32_t* values);
hipError_t iock_exclusive_scan(hipStream_t st, const uint32_t* in, int64_t n, uint32_t* out, uint32_t* scratch,
                               uint32_t round_mask);
hipError_t iock_fill_left(hipStream_t st, int64_t nkeys, const int64_t* loffs, const uint32_t* lpost,
                          const uint32_t* lslot, const uint32_t* off, void* post, int post16);
hipError_t iock_fill_queries(hipStream_t st, int n, uint32_t L, const int64_t* doff, const uint32_t* dcount,
                             const uint32_t* dslot, const uint32_t* dpos, const uint32_t* off, void* post, int post16);
hipError_t iock_sort_lists(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, void* post,
                           uint32_t L, uint32_t n, uint32_t nblocks, uint32_t* qinfo, int post16, int sorted = 0);
hipError_t iock_export_count(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, const void* post, int post16,
                             uint32_t L, const int32_t* cid, uint32_t* out_cnt);
hipError_t iock_export_fill(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, const void* post, int post16,
                            uint32_t L, const int32_t* cid, const uint32_t* out_cnt, const int64_t* out_off, uint32_t* out);
// ioc_sort.hip: the kept keys in ascending order (okeys), the offsets of their lists (soff, nslots + 1 values: the last is the
// total), every slot's offset (slot_off), the number of kept keys (n_rows)
size_t iock_export_order_temp(uint32_t nslots);
hipError_t iock_export_order(hipStream_t st, uint32_t nslots, uint32_t cap, const uint32_t* keys, const uint32_t* cnt,
                             unsigned long long* k0, unsigned long long* k1, uint32_t* v0, uint32_t* v1, unsigned long long* scnt,
                             unsigned long long* soff, void* temp, size_t temp_bytes, uint32_t* n_rows, uint32_t* okeys, int64_t* slot_off);
hipError_t iock_gather_lists(hipStream_t st, uint32_t nlists, const int64_t* src, const int64_t* dst, const uint32_t* len,
                             const uint32_t* smin, const uint32_t* spos, uint32_t* dmin, uint32_t* dpos);
hipError_t iock_pack_rows(hipStream_t st, uint32_t nslots, const uint32_t* keys, const uint32_t* off,
                          const uint32_t* cnt, const uint32_t* qinfo, void* rows);
hipError_t iock_score(hipStream_t st, int n, uint32_t L, const int64_t* off_fwd, const int64_t* off_rev,
                      const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift, const void* post,
                      uint32_t range, uint32_t keep, uint32_t* cand_key, uint32_t* cand_size, uint32_t* cand_count,
                      unsigned long long* traversed, const uint8_t* audit_valid, unsigned long long* audit_sum,
                      uint32_t* part, uint32_t* top_all, int post16, uint32_t* pmins, uint32_t* pbnd);
hipError_t iock_gap_bounds(hipStream_t st, int n, const int64_t* off_fwd, const int64_t* off_rev, const uint32_t* pos,
                           const uint32_t* hpc_len, const uint8_t* err_cell, const int32_t* glim, uint2* out, const uint32_t* min_total,
                           uint32_t keep, uint32_t* keep_q);
void iock_set_score_keep(const uint32_t* keep_q);
hipError_t iock_guess_valid(hipStream_t st, int n, const int64_t* off_fwd, const int64_t* off_rev,
                            const uint32_t* top_all, uint8_t* valid);
hipError_t iock_decide_sweep(hipStream_t st, const void* args, int nblocks, int eval_blocks, uint32_t* q_count2);
hipError_t iock_decide_phase2(hipStream_t st, const void* args, int nblocks, int eval_blocks, uint32_t* q_count2);
hipError_t iock_copy_prefix_valid(hipStream_t st, int first, const uint8_t* vin, uint8_t* vout, uint32_t* ctl = nullptr);
hipError_t iock_query_compact(hipStream_t st, const uint32_t* hist, const uint32_t* first, uint32_t n2, uint32_t cap, uint32_t* out);
hipError_t iock_query_table_many(hipStream_t st, int nq, const int32_t* qlist, uint64_t stride, uint32_t L, const int64_t* off_fwd,
                                 const int64_t* off_rev, const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift,
                                 const void* post, const uint8_t* valid, uint32_t* hist, uint32_t* first, int post16, uint32_t ccap,
                                 uint32_t* out);
hipError_t iock_query_table(hipStream_t st, int j, uint32_t L, const int64_t* off_fwd, const int64_t* off_rev,
                            const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift, const void* post,
                            const uint8_t* valid, uint32_t* hist, uint32_t* first, int post16);
size_t iock_decide_args_size();
void iock_set_score_variant(int v);
void iock_set_part32(int v);
void iock_set_score_oob(int v);
void iock_set_score_shard(int stride, int offset);
hipError_t iock_shard_mask_u8(hipStream_t st, uint8_t* a, uint8_t* b, int from, int n, int stride, int offset, uint32_t* ctl);
hipError_t iock_shard_mask_i32(hipStream_t st, int32_t* a, int n, int stride, int offset);
hipError_t iock_lds_oob_probe(hipStream_t st, uint32_t* d_result, uint32_t* h_result);

// ---- sort-stage kernels (ioc_extract.hip) ----
hipError_t iock_qual_scores(hipStream_t st, int n, const int64_t* offs, const uint8_t* qual, int k,
                            const double* tab_capped, const double* tab_nomin, double* score, double* err);
hipError_t iock_hpc(hipStream_t st, int n, const int64_t* offs, const uint8_t* seq, const uint8_t* qual,
                    uint8_t* hseq, uint8_t* hqual, uint32_t* hlen, int32_t* status);
hipError_t iock_hpc_error(hipStream_t st, int n, const int64_t* offs, const uint8_t* hqual, const uint32_t* hlen,
                          const double* tab_nomin, double* err);
hipError_t iock_minimizers(hipStream_t st, int n, const int64_t* offs, const uint8_t* hseq, const uint32_t* hlen,
                           const int32_t* status, int k, int w, int pass, const int64_t* off_fwd,
                           const int64_t* off_rev, uint32_t* cnt_fwd, uint32_t* cnt_rev, uint32_t* omin,
                           uint32_t* opos, uint32_t max_hlen);
}

// epoch cuts of the index rows (ioc_kernels.hip, index_lookup): 7 boundaries that cut the target ids into 8 equal ranges
#define IOC_EPOCHS 7
#define IOC_EPOCH_LONG 1016u
struct Epochs {
    uint32_t e[IOC_EPOCHS];
};
extern "C" Epochs iock_epoch_bounds(uint32_t L, uint32_t n);

// ---- index build without global atomics (ioc_build_sort.hip) ----
struct IocBuildSort {
    int pairs_done = 0;  // the queries' pairs are in pk_in / pv_in already (written by k_distinct_radix)
    int n;                        // queries
    uint32_t L;                   // left clusters: query j is target L + j
    const int64_t* doff;          // [n + 1] capacity offsets of the queries' distinct lists
    const uint32_t* dcount;       // [n]
    const uint32_t* dvals;
    int64_t n_left_keys, n_left_post;
    const uint32_t* lkeys;        // left MinDB (CSR): keys, offsets, postings
    const int64_t* loffs;
    const uint32_t* lpost;
    int64_t P;                    // pairs incl. the unused tails of the queries' lists: n_left_post + doff[n]
    int post16, value_bits;
    uint32_t pad_mask;            // lists are padded to whole 16-byte units
    uint32_t *pk_in, *pk_out;     // [P]
    void *pv_in, *pv_out;         // [P] postings (u16 / u32)
    uint32_t *rid, *run_slot;     // [P + 1]: run number of every pair (exclusive scan of the flags); hash slot of every run
    uint32_t *run_start;          // [P + 1]
    uint32_t *scan_scratch;       // [scan_words]: what an exclusive scan over the table's slots needs (ceil((cap + 1) / 1024) + 1)
    size_t scan_words;
    uint32_t* ctl;                // 4 words
    void* temp;
    size_t temp_bytes;
};
size_t iock_build_sort_temp_bytes(int64_t P, int post16, int value_bits);
hipError_t iock_build_sort_phase1(hipStream_t st, const IocBuildSort* a);
hipError_t iock_build_sort_phase2(hipStream_t st, const IocBuildSort* a, uint32_t R, uint32_t n_real, uint32_t* keys, uint32_t cap, uint32_t shift,
                                  uint32_t* cnt, uint32_t* off, void* post, uint32_t* qinfo, uint32_t n_targets, uint32_t* err);

#endif
