// ioc_internal.h — context layout shared by the device TU (ioc_device.hip) and the host driver
// (ioc_host.cpp).  Not part of the public C ABI (include/isonclust2_hip.h).
#ifndef IOC_INTERNAL_H
#define IOC_INTERNAL_H

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "isonclust2_hip.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// IOC_POISON=<byte>: every fresh device allocation of the library is filled with that byte (debug aid: a kernel that reads
// memory nobody wrote gives results that change with the byte; a fresh process otherwise sees zero-filled VRAM and hides it)
void ioc_poison(void* p, size_t bytes);

struct ioc_dist_state;  // ioc_dist.cpp: the context's RCCL communicator

struct ioc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;

    bool have_params = false;
    ioc_params params{};
    int32_t h_glim[225]{};
    int32_t keep = 1;  // candidates with Size < keep can never be evaluated (cluster.cpp:376-389)

    // ---- queries ----
    int32_t n = 0;
    int64_t total = 0;
    bool borrowed = false;
    std::vector<int64_t> h_off_fwd, h_off_rev, h_doff;
    const int64_t* d_off_fwd = nullptr;
    const int64_t* d_off_rev = nullptr;
    const uint32_t* d_min = nullptr;
    const uint32_t* d_pos = nullptr;
    const uint32_t* d_hpc_len = nullptr;
    const uint8_t* d_err_cell = nullptr;
    const uint32_t* d_min_total = nullptr;
    DevBuf b_off_fwd, b_off_rev, b_min, b_pos, b_hpc_len, b_err_cell, b_min_total, b_doff;
    int32_t max_fwd = 0, max_rev = 0;
    // ioc_cluster_merge (caller's arrays valid for the whole call): the index build needs the forward minimizer values
    // only, the scoring the reverse ones too, the resolve the positions — the latter two go up on a copy stream from a
    // thread of their own while the first kernels run (ioc_queries_upload, ioc_wait_uploads)
    bool defer_uploads = false;
    std::thread up_thread;
    std::atomic<int> up_stage{2};  // 1: all minimizer values are in HBM, 2: the positions too
    std::atomic<bool> up_failed{false};  // a copy of the upload thread failed (up_err holds the text): set before the stage moves on
    uint64_t query_gen = 0;  // bumped whenever the context's queries are replaced (ioc_queries_generation)
    std::string up_err;
    hipStream_t copy_stream = nullptr;
    hipStream_t side_stream = nullptr;  // the aligner's helper launch for the wrong candidates, beside the first traceback launch
    hipEvent_t ev_side[2]{};

    // ---- left state ----
    int32_t L = 0;
    int64_t n_left_keys = 0, n_left_post = 0;
    DevBuf b_left_err, b_lkeys, b_loffs, b_lpost, b_lslot;
    // left clusters' value sets (transposed MinDB), built by ioc_left_load
    DevBuf b_lset_off, b_lset_val;
    std::vector<int64_t> h_lset_off;  // host copy of the value-set offsets (ioc_index_update)

    // ---- index ----
    bool built = false;
    uint32_t cap = 0;  // power of two; slot `cap` is reserved for the key 0xFFFFFFFF
    DevBuf b_keys, b_cnt, b_off, b_fill, b_rows, b_post, b_dvals, b_dcount, b_dslot, b_scan;
    int64_t n_post = 0;
    int post16 = 0;  // postings stored as uint16_t (L + N <= 65535)

    // ---- scoring ----
    bool scored = false;
    DevBuf b_cand_key, b_cand_size, b_cand_mapped, b_cand_count, b_qinfo, b_part, b_diag, b_top_all, b_pmins, b_pbnd;
    DevBuf b_dlong;  // ioc_index_build: the long queries' values gathered / sorted (iock_distinct_long)
    DevBuf b_exp_cid, b_exp_cnt, b_exp_off, b_exp_out, b_exp_work;  // ioc_index_export: final ids, per-slot counts / offsets, compact postings
    bool have_guess = false;
    int64_t cand_capacity = 0;

    // ---- resolve ----
    bool resolved = false;
    DevBuf b_valid0, b_valid1, b_dec_target, b_dec_strand, b_flags, b_forced_t, b_forced_s, b_misc,
        b_glim, b_queue, b_cut;
    int cur_valid = 0;
    std::vector<int32_t> h_forced_t;
    std::vector<int8_t> h_forced_s;
    bool forced_dirty = false;
    std::vector<std::vector<std::pair<int32_t, int8_t>>> last_dep_set;  // ... and the (cluster, strand) candidates among which that order picks the first
    bool want_dep_sets = false;  // (ioc_cluster_consensus: run_pipeline leaves last_order_dep / last_dep_set)
    std::vector<uint8_t> last_order_dep;  // per query of the last run_pipeline: its decision hangs on the reference's hit ORDER (a tie at the top Size, or several candidates that align)
    bool forced_host_clear = false, forced_dev_clear = false;  // nothing forced in the host arrays / in what the device holds (no upload then)
    std::vector<uint32_t> h_min_total;  // host copy of d_min_total for ioc_cluster_resident's tie replays, of queries `h_min_total_gen`
    uint64_t h_min_total_gen = ~0ull;
    uint8_t* h_pin_big = nullptr;   // pinned staging for the per-call read-backs of n-sized arrays (decisions)
    size_t h_pin_big_cap = 0;

    // ---- extraction (K1) outputs ----
    DevBuf x_min, x_pos, x_off_fwd, x_off_rev, x_hpc_len, x_hseq, x_hqual;
    std::vector<int64_t> xh_offs;
    int32_t x_n = 0;
    int64_t x_total = 0;
    std::vector<int64_t> xh_off_fwd, xh_off_rev;
    std::vector<uint32_t> xh_hpc_len;
    std::vector<int32_t> xh_status;
    std::vector<uint8_t> x_keep;

    // ---- GPU alignment fallback (ioc_align_gpu.hip) ----
    DevBuf a_pool, a_pairs, a_order, a_out, a_bnd, a_lrow, a_ck, a_cko, a_ends, a_ends2, a_xflags, a_prof;
    std::vector<uint8_t> aln_other;  // per pool sequence: holds a byte other than A C G T
    std::vector<int64_t> aln_offs;
    DevBuf b_aln_t, b_aln_s, b_tie_count, b_tie_keys;
    DevBuf b_qhist, b_qfirst, b_qout, b_qlist;
    // ioc_resolve warm start: first query whose alignment verdict changed since the last resolve (n: none; -1: no
    // resolved state to start from).  Everything before it keeps its decision (it depends on earlier queries only).
    int32_t warm_first = -1;  // ioc_query_candidates: the query's hit table (kept between calls)
    std::vector<int32_t> h_aln_t;
    std::vector<int8_t> h_aln_s;
    bool aln_verdicts = false, aln_dirty = false;
    // raw sequences of the resident queries (ioc_resident_set_sequences): sahlin on ioc_cluster_resident
    std::string res_seq;
    std::vector<int64_t> res_off;
    std::vector<double> res_err;
    bool have_res_seq = false;
    bool res_pool_ready = false;  // a_pool holds exactly res_seq
    size_t aln_lds_max = 0, aln_lds_max2 = 0, aln_lds_max3 = 0;  // dynamic LDS a k_align_fwd<true/false> workgroup may reserve (residency cap)

    // ---- alignment results kept across the device passes of ioc_cluster_consensus (see AlnDriver) ----
    std::vector<uint64_t> aln_qid, aln_lid;  // sequence identity of every right entry / left representative
    std::map<std::pair<uint64_t, uint64_t>, double> aln_cache;

    // ---- ioc_index_export result of the current resolve ----
    bool exp_valid = false;   // exp_keys / exp_offs / exp_post hold the export (consensus driver; IOC_EXPORT_HOST_ORDER)
    bool exp_dev = false;     // the export is ready ON THE DEVICE: exp_nrows keys at b_exp_work + exp_o_keys, exp_nrows + 1 int64
                              // offsets at + exp_o_offs, exp_total postings in b_exp_out (copied straight into the caller's arrays)
    uint32_t exp_nrows = 0;
    uint64_t exp_total = 0;
    size_t exp_o_keys = 0, exp_o_offs = 0;
    std::vector<uint32_t> exp_keys, exp_post;
    std::vector<int64_t> exp_offs;

    // ---- instrumentation ----
    hipEvent_t ev[6]{};
    ioc_timings tm{};
    // ---- multi-GPU (ioc_dist.cpp) ----
    ioc_dist_state* dist = nullptr;
    DevBuf b_dist_min, b_dist_pos;  // the gathered representatives' minimizer lists of ioc_dist_merge (used in place as queries)
    // sharded score + resolve (ioc_set_shard): this rank owns the queries j with j % shard_world == shard_rank
    int shard_world = 1, shard_rank = 0;
    ioc_exchange_fn shard_fn = nullptr;
    void* shard_user = nullptr;
    bool scored_sharded = false;  // the candidate tables hold the owned queries only
    int shard_exchanges = 0;
    int64_t shard_aln_pairs = 0;  // pairs THIS rank aligned in sharded alignment rounds since ioc_set_shard
    DevBuf b_shard_stage;
    bool chunked_call = false;  // the last ioc_cluster_merge ran its right batch in chunks: the resident queries are the last chunk's
    double aln_verdict_thr = -1.0;  // ioc_align_set_verdict_threshold (<= 0: exact counts)
    // The aligner's corridor model (ioc_align_gpu.hip, align_v2_run): score per base of the pairs aligned so far against their summed
    // error rate, one straight line per gap-open class (setGapOpen: 2..5).  Decides how wide a couple's corridor is PLANNED — which
    // tiles are computed, never what comes out (the certificate and the re-run see to that).
    struct CorridorFit {
        double n = 0, se = 0, sr = 0, see = 0, ser = 0, srr = 0, e_lo = 1e9, e_hi = -1e9;
    };
    CorridorFit aln_fit[4];
    int32_t aln_fit_sig[3] = {0, 0, 0};  // (match, mismatch, gap_extend) the sums belong to
    uint32_t* h_pin = nullptr;     // 256 bytes of pinned host memory: the read-backs of the resolve's sweeps
    DevBuf b_bsort;                // the sorted index build's arena (ioc_build_sort.hip)
    DevBuf b_gap_bound, b_keep_q;  // k_gap_bounds' table of the current queries; the per-query compaction threshold (fast mode)
    uint64_t gap_bound_gen = ~0ull;  // query_gen the table was computed for (ioc_set_params resets it)
    bool gap_bound_cut = false;      // ... with keep_q written
    bool keep_q_on = false;          // the candidate lists of the last ioc_score were cut at b_keep_q
    std::vector<uint32_t> h_keep_q;  // host copy, fetched when a candidate table is exported
    int score_oob = 0, score_oob_probe = -1;  // k_score_part's variant and the probe behind it (ioc_ctx_create)
};

int ioc_fail(ioc_ctx* c, int code, const std::string& msg);
// sharded score + resolve (ioc_set_shard): one all-reduce through the caller's hook / a host array of words summed over ranks
int ioc_shard_exchange(ioc_ctx* c, void* d_buf, int64_t count, int kind);
int ioc_shard_sum_host(ioc_ctx* c, int32_t* words, int64_t count);
// waits until the background upload of the query arrays has reached `stage` (see ioc_ctx::up_stage); 2 also ends the thread
int ioc_wait_uploads(ioc_ctx* c, int stage);



// ioc_query_candidates for many queries at once (one launch per chunk, one synchronisation): per query the same
// lists — target, strand (+1 / -1), Size, first hitting Index, cached totalMapped (0xFFFFFFFF: not evaluated) —
// in ascending (strand +1 first, target) order.  Internal: used by the hitOrder replay of the sahlin driver.
struct IocCandTable {
    int q = 0;
    std::vector<int32_t> tg;
    std::vector<int8_t> st;
    std::vector<uint32_t> sz, fi, tm;
};
int ioc_query_candidates_many(ioc_ctx* c, const std::vector<int>& qs, std::vector<IocCandTable>& out);

// ioc_capi.cpp: queries whose minimizer arrays are already in HBM (ioc_batch_view::minimizers_on_device)
extern "C" int ioc_queries_upload_devmins(ioc_ctx* c, int32_t n, const int64_t* off_fwd, const int64_t* off_rev, const uint32_t* d_min_val,
                                          const uint32_t* d_min_pos, int64_t total, const uint32_t* hpc_len, const uint8_t* err_cell,
                                          const uint32_t* min_total);


// f(0) .. f(count - 1) on the host's cores (independent items only).  The workers are a pool that lives with the process
// (ioc_host.cpp): the consensus path comes here ~1000 times per batch, and starting 16 threads per call cost 0.4 s of it.
// A region entered while another one runs (another context on another thread, or a nested call) starts threads of its own.
#include <atomic>
#include <functional>
#include <thread>
void ioc_pool_run(size_t count, size_t nthreads, const std::function<void(size_t)>& f);
template <typename F>
static inline void ioc_parallel_for(size_t count, F f, size_t serial_below = 4)
{
    const size_t hw = std::thread::hardware_concurrency();
    const size_t nt = count < serial_below ? 1 : std::min<size_t>(count, std::max<size_t>(1, std::min<size_t>(16, hw)));
    if (nt <= 1) {
        for (size_t x = 0; x < count; ++x) f(x);
        return;
    }
    ioc_pool_run(count, nt, std::function<void(size_t)>(std::ref(f)));
}

#endif