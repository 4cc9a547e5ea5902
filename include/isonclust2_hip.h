/*
 * isonclust2_hip.h — C ABI of the MI355X-native read->cluster assignment path of isONclust2.
 *
 * The reference has no plugin/FFI interface; its seam for this path is the C++ call
 *     StrandedCluster getBestCluster(rightId, leftBatch, rightBatch, sharedMinTab)   src/cluster.h:18-20
 * made once per read inside ClusterSortedReads (src/cluster.cpp:166), plus the index mutators
 * AddMinimizers (src/cluster.cpp:180, src/minimizer.cpp:31-42) and the sort-stage feeders
 * (src/qualscore.cpp:39-136).  This header is what a C++ maintainer binds instead: plain pointers
 * and sizes, no C++/torch types, every call returns 0 or a negative ioc_status and never throws.
 *
 * One context per GPU; a context is used by one host thread at a time (the reference `cluster`
 * is single-threaded and non-reentrant: globals at src/cluster.cpp:21-23, src/minimizer.cpp:15).
 *
 * Data model on the device (all SoA, 32-bit):
 *   queries   = the right batch's clusterable entries in loop order (src/cluster.cpp:115), each with
 *               forward and reverse minimizer lists (value, position; Index == ordinal,
 *               src/minimizer.cpp:78-123), HPC length, error-rate cell and integer pass threshold;
 *   targets   = L existing left clusters (ids 0..L-1, from the persisted MinDB) followed by the
 *               queries themselves as tentative new clusters (id L+j) — with consensus off a
 *               representative never changes (src/cluster.cpp:263-265), so (Size, totalMapped) of
 *               (query j, target t) is independent of every clustering decision;
 *   index     = open-addressed hash  minimizer value -> posting list of target ids.
 */
#ifndef ISONCLUST2_HIP_H
#define ISONCLUST2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ioc_ctx ioc_ctx;

typedef enum {
    IOC_OK = 0,
    IOC_ERR_ARG = -1,       /* bad argument / shape */
    IOC_ERR_HIP = -2,       /* HIP runtime error (ioc_last_error has the text) */
    IOC_ERR_STATE = -3,     /* call order violated */
    IOC_ERR_CAPACITY = -4,  /* a documented device-side limit was exceeded */
    IOC_ERR_TABLE = -5,     /* empirical probability table lookup failure (p_emp_prob.cpp:87-89) */
    IOC_ERR_NO_DEVICE = -6,
    IOC_ERR_INPUT = -7      /* input the reference would exit(1)/throw on */
} ioc_status;

/* ClsMode, src/args.h:7 */
enum { IOC_MODE_SAHLIN = 0, IOC_MODE_FAST = 1, IOC_MODE_FURIOUS = 2, IOC_MODE_NONE = 3 };

/* The clustering parameters frozen at sort time (CmdArgs, src/args.h:9-37) that the path reads
 * (src/cluster.cpp:366-369, 534-536). */
typedef struct {
    int32_t k;                /* KmerSize */
    int32_t w;                /* WindowSize */
    int32_t min_shared;       /* MinShared */
    int32_t mode;             /* IOC_MODE_* */
    double min_fraction;      /* MinFraction */
    double mapped_threshold;  /* MappedThreshold */
    double min_prob_no_hits;  /* MinProbNoHits */
    double aligned_threshold; /* AlignedThreshold (host fallback only) */
} ioc_params;

/* ---- context -------------------------------------------------------------------------------- */
int ioc_ctx_create(int device, ioc_ctx** out);
void ioc_ctx_destroy(ioc_ctx* ctx);
/* Gives the aligner's scratch (the checkpoint arena: gigabytes after an alignment-mode batch) back to the driver; the next call
 * that needs it allocates again.  A one-shot process calls this as soon as its clustering call has returned: the driver wipes
 * released VRAM before it hands it out again (8 GB: ~0.4 s), and a process that starts right after this one would otherwise
 * wait for that inside its own first large hipMalloc (profiles/r05_cli_breakdown.txt). */
int ioc_ctx_trim(ioc_ctx* ctx);
/* Optional, for a process that runs ONE batch: has the code objects of the clustering path (and of the aligner, if
 * alignment_mode != 0) loaded by background threads now, beside the caller's own preparation, instead of at each file's first
 * launch.  Returns at once; nothing depends on it. */
int ioc_ctx_prewarm(ioc_ctx* ctx, int32_t alignment_mode);
const char* ioc_last_error(const ioc_ctx* ctx);
/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL = the context's own stream. */
int ioc_set_stream(ioc_ctx* ctx, void* hip_stream);
int ioc_synchronize(ioc_ctx* ctx);

/* Parameters + the 15x15 integer gap-limit table, gap_limit[e_cl-1][e_rd-1] =
 * max{ n : pow(1 - P(e_cl, e_rd), n) >= MinProbNoHits }: the monotone predicate of
 * getMappedRatio (src/cluster.cpp:333-347) evaluated ONCE on the host with the host libm,
 * compared as integers on the device.  ioc_host_gap_limits() fills it from the table file. */
int ioc_set_params(ioc_ctx* ctx, const ioc_params* p, const int32_t gap_limit[225]);

/* ---- queries (right batch) ------------------------------------------------------------------- */
/* Host -> device upload of the query SoA.  off_fwd/off_rev: [n+1] element offsets into
 * min_val/min_pos (total = number of minimizers, fwd and rev, of all queries; the positions of a list ascend with the index,
 * as the extractor's do: getMappedRatio subtracts consecutive ones as unsigned numbers).  err_cell in 1..15 =
 * clamp(round(100 * HpcSeq.ErrorRate)) (src/p_emp_prob.cpp:66-84).  min_total[j] = smallest integer
 * T with float(double(T)/double(hpc_len[j])) >= MappedThreshold (src/cluster.cpp:390-400). */
int ioc_queries_upload(ioc_ctx* ctx, int32_t n, const int64_t* off_fwd, const int64_t* off_rev,
                       const uint32_t* min_val, const uint32_t* min_pos, int64_t total,
                       const uint32_t* hpc_len, const uint8_t* err_cell, const uint32_t* min_total);
/* Same, with every pointer already a device pointer on ctx's device (inputs resident in HBM,
 * e.g. produced by ioc_extract_minimizers); the context borrows them until the next upload/bind. */
int ioc_queries_bind_device(ioc_ctx* ctx, int32_t n, const int64_t* d_off_fwd, const int64_t* d_off_rev,
                            const uint32_t* d_min_val, const uint32_t* d_min_pos, int64_t total,
                            const uint32_t* d_hpc_len, const uint8_t* d_err_cell,
                            const uint32_t* d_min_total, const int64_t* h_off_fwd,
                            const int64_t* h_off_rev);

/* ---- left state (merge: `cluster -l L -r R`, src/main.cpp:247-261) ----------------------------- */
/* The persisted MinDB (src/minimizer.h:60-61) as CSR: n_keys distinct values, offs[n_keys+1],
 * postings = ascending cluster ids < n_clusters.  n_clusters = 0 resets to initial clustering. */
int ioc_left_load(ioc_ctx* ctx, int32_t n_clusters, const uint8_t* cls_err_cell, int64_t n_keys,
                  const uint32_t* keys, const int64_t* offs, const uint32_t* postings);

/* UpdateMinDB (src/minimizer.cpp:124-160, called at src/cluster.cpp:296 once a cluster's consensus replaced its
 * representative): left cluster `cls` leaves the posting lists of the values only old_min has and enters, at
 * its sorted place, the lists of the values only new_min has (both arrays: the representative's forward
 * minimizer VALUES, duplicates allowed).  Lists that become empty stay as keys, unseen values open new keys
 * (`db[m]`, :146, :156).  old_min must be what the index holds for the cluster — in the reference it always
 * is (AddMinimizers / the previous UpdateMinDB put it there) — otherwise IOC_ERR_INPUT.  new_err_cell (1..15,
 * 0 = unchanged) is the cell of the representative's re-weighted HPC error rate (src/consensus.cpp:56-58, 104).
 * Device-side rewrite of the CSR; the combined index must be rebuilt afterwards (ioc_index_build). */
int ioc_index_update(ioc_ctx* ctx, int32_t cls, const uint32_t* old_min, int64_t n_old, const uint32_t* new_min,
                     int64_t n_new, uint8_t new_err_cell);
/* The left MinDB as it stands on the device (after ioc_left_load / ioc_index_update), CSR with ascending keys,
 * keys with empty lists included.  Call with keys == NULL to size. */
int ioc_left_export(ioc_ctx* ctx, int64_t* n_keys, int64_t* n_postings, uint32_t* keys, int64_t* offs,
                    uint32_t* postings);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* AddMinimizers for every tentative representative at once (src/minimizer.cpp:31-42): per-query
 * sorted distinct forward values, hash insert, posting lists. */
int ioc_index_build(ioc_ctx* ctx);
/* GetMinimizerHits + ConsolidateMinimizerHits + the Size part of SortMinimizerHits
 * (src/minimizer.cpp:44-76, src/cluster.cpp:609-636) for every query against every earlier
 * target: per-query candidate lists (target, strand, Size). */
int ioc_score(ioc_ctx* ctx);
/* getBestClusterMapping + getMappedRatio (src/cluster.cpp:324-406) for all queries, iterated to the
 * unique fixed point of the greedy loop (src/cluster.cpp:115-310).  Queries listed in
 * forced_* (host decisions: alignment fallback results in sahlin mode) are taken as given.
 * n_iter (optional) receives the number of parallel sweeps. */
int ioc_resolve(ioc_ctx* ctx, int32_t* n_iter);
/* Per query: target >= 0 joined target id (left cluster id, or L + index of the query that opened
 * the cluster), -1 = opens a new cluster; strand +1/-1 (0 for new); flags bit0 = order-dependent tie
 * (>= 2 passing candidates at the winning Size: host replays the libstdc++ order), bit1 = no
 * mapping hit but top >= MinShared (sahlin/furious: host alignment fallback, cluster.cpp:553-566). */
int ioc_get_decisions(ioc_ctx* ctx, int32_t* target, int8_t* strand, uint8_t* flags);
/* Per query, the Size below which the mapping walk never looks at a candidate: int(top * MinFraction) with
 * top = the largest Size among the current clusters (cluster.cpp:381-403), or INT32_MAX when the query has no
 * walk at all (top < MinShared, or a forced decision).  Valid after ioc_resolve. */
int ioc_get_cuts(ioc_ctx* ctx, int32_t* cut);
/* Host override of one query's decision (tie replay / alignment fallback); takes effect in the
 * next ioc_resolve. target -1 = new cluster. */
int ioc_force_decision(ioc_ctx* ctx, int32_t query, int32_t target, int32_t strand);
int ioc_clear_forced(ioc_ctx* ctx);
/* Verdicts of the alignment fallback (getBestClusterAln, src/cluster.cpp:461-515), one per query:
 * target[q] = INT32_MIN (none yet), -1 (no candidate aligned: opens a cluster) or the joined target with
 * strand[q] = +1/-1.  Unlike a forced decision a verdict is used by ioc_resolve only when the query's
 * mapping walk finds nothing although top >= MinShared (flags bit1 stays set), so verdicts may be
 * supplied speculatively for many queries at once.  target == NULL switches verdicts off. */
int ioc_set_aln_verdicts(ioc_ctx* ctx, const int32_t* target, const int8_t* strand);
/* With verdicts set, ioc_resolve also reports per query the candidates tied at the top Size among the
 * current clusters — the ones getBestClusterAln tries (cluster.cpp:481-489): count[n] and up to
 * IOC_TIE_SLOTS keys per query (keys[IOC_TIE_SLOTS * q + i] = target << 1 | (strand == -1), unordered);
 * more than that: ioc_query_candidates. */
#define IOC_TIE_SLOTS 16
int ioc_get_ties(ioc_ctx* ctx, uint32_t* count, uint32_t* keys);

/* The candidate list the SCORING kernels wrote for query q (before any resolve): key = target << 1 | strand (strand bit
 * 1 = reverse), size = Size, for every (target, strand) with Size >= int(MinShared * MinFraction), in target order — in fast
 * mode: with Size >= the smallest Size that can pass the query's mapped-fraction test at all (an upper bound of totalMapped
 * from the query's minimizer positions and gap limits; IOC_SCORE_KEEPQ=0 keeps the uniform threshold).
 * Returns the count (copies at most cap).  Test instrument: lets the two builds of the scoring kernel (with / without the
 * per-posting window test, ioc_timings::score_oob) be compared histogram by histogram. */
int ioc_scored_candidates(ioc_ctx* ctx, int32_t q, int32_t cap, uint32_t* key, uint32_t* size);
/* Full candidate table of one query against the targets that are clusters under the current
 * decisions, in the fields the reference's hit map holds (src/minimizer.cpp:44-76): target id,
 * strand, Size, Index of the first hitting read minimizer, and totalMapped (0xFFFFFFFF if not
 * evaluated; 0xFFFFFFFE if the candidate was rejected without an evaluation because an upper bound of its totalMapped —
 * (Size - 1) x the widest span of a passing gap + the widest head and tail — is below the query's threshold: it fails).
 * Returns the count (<= cap) or a negative status. */
int ioc_query_candidates(ioc_ctx* ctx, int32_t query, int32_t cap, int32_t* target, int8_t* strand,
                         uint32_t* size, uint32_t* first_index, uint32_t* total_mapped);

/* MinDB after clustering (AddMinimizers applied for every query that opened a cluster): CSR with
 * final cluster ids, keys ascending.  Call with keys == NULL to size (n_keys, n_postings): the CSR is
 * built and kept on the device then, and the call with the arrays copies it straight into them. */
int ioc_index_export(ioc_ctx* ctx, int64_t* n_keys, int64_t* n_postings, uint32_t* keys,
                     int64_t* offs, uint32_t* postings);

/* ---- sort-stage feeders (src/qualscore.cpp:39-136, src/hpc.cpp:4-32, src/kmer_index.cpp:5-17,
 *      src/minimizer.cpp:78-123) ------------------------------------------------------------------ */
/* CalcQualScore / CalcErrorRate for n reads (host pointers; offs[n+1] into qual).  score[i] < 0
 * mirrors FillQualScores' -1 (src/qualscore.cpp:22-34). */
int ioc_qual_scores(ioc_ctx* ctx, int32_t n, const int64_t* offs, const uint8_t* qual, int32_t k,
                    double* score, double* err_rate);
/* HomopolymerCompress + RevComp + KmerEncodeSeq x2 + GetKmerMinimizers x2 + CalcErrorRate(hpc quals)
 * for n reads.  Outputs stay on the device in the layout ioc_queries_bind_device takes; the host
 * receives lengths/offsets/error rates.  status[i]: 0 ok, 1 HPC length < 2k or < w
 * (src/qualscore.cpp:65-73), 2 non-ACGT base (RevComp throws, src/util.cpp:31-33). */
int ioc_extract_minimizers(ioc_ctx* ctx, int32_t n, const int64_t* offs, const uint8_t* seq,
                           const uint8_t* qual, int32_t k, int32_t w, uint32_t* hpc_len,
                           double* hpc_err, int64_t* off_fwd, int64_t* off_rev, int32_t* status);
/* Device->host copy of the HPC sequences / qualities of the last ioc_extract_minimizers (ASCII; read
 * i occupies [offs[i], offs[i] + hpc_len[i]) of the buffers, which have the size of the raw input). */
int ioc_extracted_hpc_download(ioc_ctx* ctx, char* hpc_seq, char* hpc_qual, int64_t cap);
/* Device->host copy of the minimizers produced by the last ioc_extract_minimizers. */
int ioc_extracted_download(ioc_ctx* ctx, uint32_t* min_val, uint32_t* min_pos, int64_t cap);
/* Make the extracted minimizers the current queries (no host round trip of the 8 B/minimizer SoA).
 * keep[i] != 0 selects read i (gates of src/cluster.cpp:116-160 applied by the caller). */
int ioc_queries_from_extracted(ioc_ctx* ctx, const uint8_t* keep, const uint8_t* err_cell,
                               const uint32_t* min_total);

/* ---- instrumentation ------------------------------------------------------------------------ */
typedef struct {
    float ms_build;    /* ioc_index_build, HIP events on the launch stream */
    float ms_score;    /* ioc_score (the dominant kernel)                  */
    float ms_resolve;  /* ioc_resolve                                      */
    int32_t resolve_iters;
    int32_t n_queries;
    int64_t n_minimizers;      /* probes M                                  */
    int64_t n_index_postings;  /* postings stored in the device index       */
    int64_t n_candidates;      /* candidate entries written by ioc_score    */
    int64_t n_mapped_evals;    /* (query,target,strand) mapped-ratio evaluations */
    int64_t postings_traversed;/* postings read by ioc_score (if counted)   */
    /* GPU alignment fallback, summed over the ioc_align_pairs calls since the last ioc_index_build
     * (HIP events around the two passes on the launch stream) */
    float ms_align_fwd;        /* k_align_fwd: score-only DP + checkpoints   */
    float ms_align_trace;      /* k_align_trace: tiled traceback + windows   */
    int64_t n_align_pairs;
    int64_t n_align_cells;     /* sum of query length x reference length     */
    int64_t n_align_refused;   /* pairs the packed 16-bit forward pass handed to the 32-bit one */
    /* the scoring kernel's variant (decided once, in ioc_ctx_create): 1 = counters at the end of the LDS allocation, no
     * window test (the hardware's bounds check drops what the test would reject), 0 = window test per posting.
     * score_oob_probe: result of the context's run-time probe of that hardware behaviour — 0 passed, > 0 failed (bit 0 a
     * word of some workgroup's LDS changed, bit 1 an in-bounds counter lost an add, bit 2 the probe did not finish),
     * -1 not run (IOC_SCORE_OOB=0 / 1 forces the variant).  A failed probe selects variant 0. */
    int32_t score_oob;
    int32_t score_oob_probe;
    /* the aligner's checkpoint arena as the last ioc_align_pairs call sized it (bytes), the launches (slices) it took, and the
     * version that ran: 2 = two pairs per wave, coarse checkpoints (the default), 1 = one pair per workgroup, fine checkpoints
     * (IOC_ALIGN_ARENA=fat, pairs with letters other than A C G T, pairs the 16-bit window of version 2 refused) */
    int64_t align_arena_bytes;
    int32_t align_slices;
    int32_t align_version;
    /* cells of the DP matrices the forward pass really computed (version 2 leaves out the tiles outside its certified corridor;
     * n_align_cells stays the reference's figure: every pair's whole matrix) */
    int64_t n_align_cells_computed;
} ioc_timings;
int ioc_get_timings(ioc_ctx* ctx, ioc_timings* out);
/* Instrumentation (one extra scoring launch, outside any timed region): the number of postings the
 * reference's GetMinimizerHits would traverse on this batch = sum over queries of Size over the
 * targets that are clusters — the H of the algorithmic-bytes figure. */
int ioc_count_reference_postings(ioc_ctx* ctx, int64_t* n_postings);

/* ---- host-side helpers (pure host code, no GPU needed) ----------------------------------------- */
/* Fill gap_limit[225] and p_shared[225] for (k, w) from the table file (isonclust2_amd/data/
 * pmin_shared.bin; rows selected as src/p_emp_prob.cpp:22-47).  Returns IOC_ERR_TABLE if no row
 * matches (k outside 10..30 etc.). */
int ioc_host_gap_limits(const char* table_path, int32_t k, int32_t w, double min_prob_no_hits,
                        int32_t* gap_limit, double* p_shared);
/* clamp(round(100*e)) in 1..15, src/p_emp_prob.cpp:66-84 + src/util.cpp:6-10 */
uint8_t ioc_host_err_cell(double err_rate);
/* smallest T with float(double(T)/double(hpc_len)) >= mapped_threshold (src/cluster.cpp:390-400) */
uint32_t ioc_host_min_total(uint32_t hpc_len, double mapped_threshold);

/* ---- host alignment fallback (sahlin / furious; stays on the host, src/cluster.cpp:408-515) ------ */
/* Semi-global affine alignment replacing parasail_sg_trace_scan_16/32 + parasail_result_get_traceback
 * (src/cluster.cpp:413-419, 500-502): all four ends free, gap of length n costs open + (n-1)*extend;
 * comp receives the comparison string of the whole alignment (0x7C = identical bases, ' ' otherwise).
 * Returns its length (comp_cap >= qlen + rlen + 1).  Parity with parasail's tie-breaking is pinned only
 * by the reference's AlnRatioTest vector. */
int ioc_host_align(const char* query, int32_t qlen, const char* ref, int32_t rlen, int32_t match,
                   int32_t mismatch, int32_t gap_open, int32_t gap_extend, char* comp, int32_t comp_cap,
                   int32_t* score_out);
int32_t ioc_host_gap_open(double e1_plus_e2);                     /* setGapOpen,  src/cluster.cpp:425-440 */
double ioc_host_aln_ratio(const char* comp, int32_t comp_len, double e, uint32_t slen, uint32_t k);
                                                                  /* getAlnRatio, src/cluster.cpp:442-459 */

/* ---- the same alignment on the GPU, batched (getBestClusterAln's ParasailAlign + getAlnRatio,
 * src/cluster.cpp:408-423, 442-459, 498-507) ------------------------------------------------------- */
/* One pair = read `query` against representative `ref` (indices into the sequence pool), the reference
 * optionally reverse-complemented (RevComp, src/util.cpp:13-38; cluster.cpp:491-495); e = sum of the two
 * raw error rates (cluster.cpp:497), which selects the gap-open penalty (setGapOpen) and the per-window
 * match limit floor((1-e)k).  Results are bit-identical to ioc_host_align + ioc_host_aln_ratio. */
typedef struct {
    int32_t query;
    int32_t ref;
    int32_t ref_revcomp;
    int32_t reserved;   /* 0, or a HINT > 0: any measure of how much alike the caller expects the two sequences to be that can be
                         * compared across the pairs of one call (the cluster drivers pass the query's cut, i.e. its top count of
                         * shared minimizers).  The aligner schedules pairs hinted far below the call's median together; no result
                         * depends on it. */
    double e;
} ioc_aln_pair;
/* Upload the sequence pool (concatenated raw sequences, offs[n_seqs + 1], offs[0] == 0). */
int ioc_align_set_pool(ioc_ctx* ctx, int32_t n_seqs, const char* seqs, const int64_t* offs);
/* Align n_pairs pairs; any of the outputs may be NULL.  out_windows = number of qualifying k-windows
 * (`aligned` of getAlnRatio), out_ratio = out_windows / query length (its return value). */
int ioc_align_pairs(ioc_ctx* ctx, int32_t n_pairs, const ioc_aln_pair* pairs, int32_t k, int32_t match,
                    int32_t mismatch, int32_t gap_extend, int32_t* out_score, int64_t* out_windows,
                    double* out_ratio);
/* Verdict mode.  The clustering loop only ever asks whether out_ratio >= AlignedThreshold (src/cluster.cpp:503).  With a
 * threshold > 0 set here, the traceback of a pair may stop as soon as that comparison is decided — the count of good windows
 * has reached the smallest count whose ratio passes (what is still to come can only add), or can no longer reach it (every
 * further column of the alignment ends at most one more window) — and out_windows / out_ratio are then lower bounds that
 * compare with the threshold exactly as the full counts would (out_score is always exact).  <= 0 (the default): exact
 * counts.  The cluster drivers (ioc_cluster_batch / merge / consensus) switch it on around their own alignment batches
 * unless IOC_ALIGN_VERDICT=0. */
int ioc_align_set_verdict_threshold(ioc_ctx* ctx, double aligned_threshold);

/* The minimizer lists of the entries `entries[0..n_idx)` of the context's CURRENT queries (ioc_queries_upload /
 * ioc_queries_from_extracted / the batch of the last ioc_cluster_batch), gathered on the device into caller-owned
 * DEVICE buffers in the compact layout of ioc_batch_view (all forward lists, then all reverse lists): what a rank
 * contributes to the merge's all-gather — the representatives' records never leave HBM (src/cluster.cpp:537: a right
 * cluster is matched through its representative's Mins / RevMins).  off_fwd / off_rev [n_idx + 1] (host) receive the
 * offsets; cap = capacity of the two buffers in words.  Returns the number of words written or a negative status. */
/* Generation of the context's queries: changes whenever they are replaced (ioc_queries_upload, ioc_queries_bind_device,
 * ioc_queries_from_extracted, every ioc_cluster_* driver call).  A caller that keeps entry numbers of a batch for a later
 * ioc_gather_records_device compares the value it saw after the clustering call. */
int64_t ioc_queries_generation(const ioc_ctx* ctx);
int64_t ioc_gather_records_device(ioc_ctx* ctx, int32_t n_idx, const int32_t* entries, uint32_t* d_out_min,
                                  uint32_t* d_out_pos, int64_t cap, int64_t* off_fwd, int64_t* off_rev);

/* ---- host driver: ClusterSortedReads on flat arrays (src/cluster.cpp:67-322, consensus off) ------ */
typedef struct {
    /* right batch, one record per entry in loop order */
    int32_t n;
    const int64_t* off_fwd;    /* [n+1] */
    const int64_t* off_rev;    /* [n+1] */
    const uint32_t* min_val;
    const uint32_t* min_pos;
    int64_t total;
    const uint32_t* raw_len;   /* RawSeq length        */
    const uint32_t* hpc_len;   /* HpcSeq length        */
    const double* score;       /* RawSeq Score()       */
    const double* raw_err;     /* RawSeq ErrorRate()   */
    const double* hpc_err;     /* HpcSeq ErrorRate()   */
    const uint8_t* state;      /* 0 clusterable, 1 null placeholder (skipped, cluster.cpp:125) */
    double min_qual;           /* CmdArgs::MinQual     */
    /* merge only (right batch already clustered: one record per right cluster, the arrays above
     * describe its representative, cluster.cpp:537): */
    /* sahlin / furious only: raw sequences for the host alignment fallback (cluster.cpp:461-515) */
    const char* raw_seq;       /* RawSeq->Str() of all entries, concatenated; NULL in fast mode */
    const int64_t* raw_off;    /* [n+1] */
    const int32_t* n_members;  /* reads[i]->size() - 1; NULL = fresh reads */
    int32_t depth;             /* right Batch::Depth; the MinClsSize filter applies when > 0 (:119-123) */
    int32_t min_cls_size;      /* left SortArgs.MinClsSize after the -A override (main.cpp:329-331) */
    /* several clustered batches merged in ONE pass (the left fold ((b0 + b1) + b2) ... of freshly clustered batches makes
     * the decisions of one loop over b0's, b1's, ... representatives in that order: cluster.cpp:178-217 only appends):
     * entries with is_cluster[i] != 0 are the leftmost batch's clusters — they keep their cluster (ids in entry order)
     * without being matched, exactly as if they had come in through ioc_left_view.  NULL: none. */
    const uint8_t* is_cluster;
    /* != 0: min_val / min_pos are DEVICE pointers (e.g. the output of ioc_gather_records_device after an RCCL
     * all-gather); they are used in place and must stay valid until the context's queries are replaced.  Entries
     * that the gates skip must then carry no minimizers. */
    int32_t minimizers_on_device;
} ioc_batch_view;

/* Left batch of a merge (`cluster -l L -r R`): its clusters' representative HPC error rates and the
 * persisted MinDB as CSR (keys strictly ascending, posting lists strictly ascending cluster ids). */
typedef struct {
    int32_t n_clusters;
    const double* cls_hpc_err; /* [n_clusters] HpcSeq->ErrorRate() of each representative */
    int64_t n_keys;            /* -1 with keys == NULL: keep the left state that is on the device
                                  (ioc_left_load, then any ioc_index_update); n_clusters must match it */
    const uint32_t* keys;
    const int64_t* offs;       /* [n_keys+1] */
    const uint32_t* postings;
    /* sahlin / furious only: the representatives' raw sequences and raw error rates */
    const char* rep_seq;
    const int64_t* rep_off;    /* [n_clusters+1] */
    const double* cls_raw_err; /* RawSeq->ErrorRate() of each representative */
} ioc_left_view;

typedef struct {
    int64_t n_clusters;
    int64_t n_joined;
    int64_t n_gated;
    int64_t n_tie_replays;
    int64_t n_aln_invoked;   /* reads that reached the alignment fallback (ALN_INVOKED, cluster.cpp:21,559) */
    int32_t resolve_iters;
    int32_t aln_rounds;      /* verdict rounds of the alignment fallback driver */
    int64_t n_aln_pairs;     /* (read, candidate, strand) pairs actually aligned */
    int64_t n_aln_order_dep; /* verdicts that depended on the reference's candidate order */
    int64_t n_cons_invoked;  /* representatives replaced by a consensus (CONS_INVOKED, cluster.cpp:22,295) */
    int64_t n_cons_restarts; /* device passes of ioc_cluster_consensus (one + one per consensus event) */
} ioc_cluster_stats;

/* Initial clustering of one sorted batch (`cluster -l batch.cer`, src/main.cpp:262-275):
 * out_cls[i] = final cluster id of entry i (-1 if gated), out_strand[i] = MatchStrand after the
 * flips of src/cluster.cpp:235-246 (+1 for entries that open a cluster). */
int ioc_cluster_batch(ioc_ctx* ctx, const ioc_params* p, const char* table_path,
                      const ioc_batch_view* right, int32_t* out_cls, int8_t* out_strand,
                      ioc_cluster_stats* stats);
/* Merge (`cluster -l L -r R`, src/cluster.cpp:67-322 with both batches real): every right cluster's
 * representative is a query against the left MinDB; out_cls[i] = left cluster the right cluster i
 * ends up in (existing id, or a new id >= left->n_clusters in creation order, or -1 if filtered),
 * out_strand[i] = +1 / -1 (-1: every member's MatchStrand flips, :235-246).  left == NULL is
 * ioc_cluster_batch.  ioc_index_export afterwards returns the merged MinDB.  (The arrays of `right`
 * are read until the call returns: a large batch's reverse lists and positions are still being
 * uploaded, by a thread of the library, while the first kernels run.)
 * A right batch of more than 131 072 entries (one device pass: the all-pairs candidate tables grow with the square of the
 * entries) runs in chunks of that many, in order, each against the left state the chunks before it left behind — the
 * reference's one loop; the results are those of one pass.  After such a call the context's resident queries are the last
 * chunk's (ioc_gather_records_device refuses).  No limit on a read's length: queries of more than 8192 forward minimizers
 * are sorted in global memory instead of LDS (src/minimizer.cpp:78-123 has plain vectors). */
int ioc_cluster_merge(ioc_ctx* ctx, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                      const ioc_batch_view* right, int32_t* out_cls, int8_t* out_strand,
                      ioc_cluster_stats* stats);
/* ---- consensus mode (ConsMaxSize > 0; src/cluster.cpp:200-204, 263-309, src/consensus.cpp) -------------------- */
/* The partial-order graphs of the consensus (spoa in the reference) stay with the caller, behind the operations
 * the reference performs on them.  side 0 = leftBatch->ConsGs[idx] (idx = left cluster id), side 1 =
 * rightBatch->ConsGs[idx] (idx = right entry; merges).  Every callback returns >= 0, or < 0 to abort. */
typedef struct {
    const char* raw_seq;      /* the consensus = new RawSeq->Str() of the representative (consensus.cpp:93) */
    int32_t raw_len;
    char raw_qual;            /* every quality character of it (consensus.cpp:98-100) */
    double raw_err;           /* RawSeq->ErrorRate(): weighted mean (consensus.cpp:60-62) */
    double raw_score;         /* RawSeq->Score() = raw_err * length (consensus.cpp:96) */
    const char* hpc_seq;      /* HomopolymerCompress of the consensus */
    int32_t hpc_len;
    double hpc_err;           /* HpcSeq->ErrorRate(): weighted mean (consensus.cpp:56-58, 121) */
    const uint32_t* fwd_min;  /* Mins (Index = ordinal), consensus.cpp:123 */
    const uint32_t* fwd_pos;
    int32_t n_fwd;
    const uint32_t* rev_min;  /* RevMins, consensus.cpp:124 */
    const uint32_t* rev_pos;
    int32_t n_rev;
    int32_t entry;            /* right entry whose join triggered it (the name is cons_<BatchNr>_<entry>, cluster.cpp:280-282) */
} ioc_rep_record;
typedef struct {
    void* user;
    int (*create)(void* user, int side, int idx, const char* seq, int len);                 /* new graph + AddSeqToGraph(seq, 1), cluster.cpp:200-204 */
    int (*size)(void* user, int side, int idx);                                             /* sequences().size(); < 0: no such graph */
    int (*add)(void* user, int side, int idx, const char* seq, int len, unsigned weight);   /* AddSeqToGraph, consensus.cpp:76-81 */
    int (*consensus)(void* user, int side, int idx, char* out, int cap);                    /* GenerateConsensus -> length, consensus.cpp:87 */
    int (*purge)(void* user, int side, int idx, const char* seq, int len, unsigned weight); /* ConsPurge, consensus.cpp:128-137 */
    void (*rep_changed)(void* user, int32_t cls, const ioc_rep_record* rec);                /* optional: the pointers die with the call */
    /* OPTIONAL (NULL: every consensus is taken at once, as the reference does): a graph store that can DEFER consensus
     * requests and undo tagged operations lets the driver collect the consensus events of many clusters of a pass and
     * have their graph additions aligned in ONE batch (the reference's result is unchanged: the driver verifies, once
     * the new representatives are known, that no entry it walked past them could see one, and rolls back otherwise). */
    const struct ioc_consensus_spec_ops* spec;
} ioc_consensus_ops;
/* tag = the right-batch entry that caused the operation (ascending over a call of ioc_cluster_consensus) */
typedef struct ioc_consensus_spec_ops {
    int (*create_tagged)(void* user, int side, int idx, const char* seq, int len, int tag);
    int (*add_tagged)(void* user, int side, int idx, const char* seq, int len, unsigned weight, int tag);
    /* GenerateConsensus of graph (side, idx) as it is after the additions queued so far: computed by flush(), fetched
     * by collect() (returns the length) */
    int (*consensus_deferred)(void* user, int side, int idx, int tag);
    /* safe_tag: operations tagged below it are never rolled back — the store may work them off in the same batches */
    int (*flush)(void* user, int safe_tag);
    int (*collect)(void* user, int side, int idx, int tag, char* out, int cap);
    /* undo every operation tagged >= first_tag (graphs, queued additions, deferred results); commit: forget how to */
    int (*rollback)(void* user, int first_tag);
    int (*commit)(void* user);
} ioc_consensus_spec_ops;
typedef struct {
    int32_t cons_min_size;      /* CmdArgs::ConsMinSize */
    int32_t cons_max_size;      /* CmdArgs::ConsMaxSize; <= 0: no consensus (graphs are still created, as in the reference) */
    int32_t cons_period;        /* CmdArgs::ConsPeriod */
    int32_t left_depth;         /* leftBatch->Depth on entry: -1 for a freshly sorted batch (qualscore.cpp:102) */
    const int32_t* left_sizes;  /* cls[c]->size() of the left clusters (ConsPeriod test, cluster.cpp:267-271); NULL = 2 each */
} ioc_consensus_args;
/* ClusterSortedReads with the consensus branch: same inputs and outputs as ioc_cluster_merge (left host MinDB
 * required, right raw sequences required).  The device pipeline runs over all remaining entries; the host
 * walks the decisions in the reference's order up to the first join that replaces a representative
 * (UpdateClusterConsensus), re-minimizes the consensus on the GPU, applies UpdateMinDB and resumes behind that
 * entry.  ioc_index_export afterwards returns the final MinDB (emptied lists included, minimizer.cpp:150-152). */
int ioc_cluster_consensus(ioc_ctx* ctx, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                          const ioc_batch_view* right, const ioc_consensus_args* args, const ioc_consensus_ops* ops,
                          int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats);

/* ---- a POA engine of its own behind those five operations (spoa is absent from the reference tree: parity with
 * spoa's alignments and consensus tie-breaks is unpinned).  Graphs and heaviest-bundle consensus on the host, the
 * sequence-to-graph DP (local alignment, convex gaps as two affine pieces, src/main.cpp:285-324: m 4, n -8, g -8, e -4,
 * q -20, c -1) on the GPU.  Limits: sequences <= 29 999 bases, <= 127 predecessors per node. ------------------- */
typedef struct ioc_poa ioc_poa;
int ioc_poa_create(ioc_ctx* ctx, int32_t m, int32_t n, int32_t g, int32_t e, int32_t q, int32_t c, ioc_poa** out);
void ioc_poa_destroy(ioc_poa* poa);
/* fills user + create / size / add / consensus / purge + spec of *ops (rep_changed is left to the caller) */
void ioc_poa_bind(ioc_poa* poa, ioc_consensus_ops* ops);
/* inspection: a graph's nodes (letter, topological order) and weighted edges; the alignment (node id or -1,
 * position or -1, forward order) and score of the last `add`.  Returns the number of pairs. */
int ioc_poa_graph_export(ioc_poa* poa, int side, int idx, int32_t* n_nodes, int32_t* n_edges, char* bases, int32_t* rank,
                         int32_t* edge_from, int32_t* edge_to, int64_t* edge_w);
int ioc_poa_last_alignment(ioc_poa* poa, int32_t cap, int32_t* nodes, int32_t* pos, int32_t* score);
/* persistence across `.cer` files (one graph per cluster, src/serialize.h:21,37; the blob layout is this build's
 * own): save returns the size (out == NULL to size), load replaces graph (side, idx). */
int64_t ioc_poa_graph_save(ioc_poa* poa, int side, int idx, uint8_t* out, int64_t cap);
int ioc_poa_graph_load(ioc_poa* poa, int side, int idx, const uint8_t* in, int64_t len);
/* count graphs at once (idx[x] <- the blob in[x] of len[x] bytes), parsed on the host's cores; all or none. */
int ioc_poa_graph_load_many(ioc_poa* poa, int side, int32_t count, const int32_t* idx, const uint8_t* const* in, const int64_t* len);

/* The same pipeline on queries already resident on the device (bench: inputs in HBM). n entries
 * must all be clusterable.  Fast mode, or sahlin mode after ioc_resident_set_sequences. */
int ioc_cluster_resident(ioc_ctx* ctx, int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats);
/* Raw sequences (RawSeq->Str(), concatenated; raw_off[n + 1]) and raw error rates of the resident
 * queries, for the alignment fallback of sahlin mode (cluster.cpp:491-497). */
int ioc_resident_set_sequences(ioc_ctx* ctx, const char* raw_seq, const int64_t* raw_off, const double* raw_err);

/* ---- multi-GPU: the merge's exchange step over RCCL; one process and one context per GPU -------------
 * Initial clustering shards batches one per GPU and needs no communication (the reference pipeline runs one `cluster`
 * process per batch, README.md:105-117).  The merge (`cluster -l L -r R`, src/cluster.cpp:67-322 with two batches) matches a
 * right cluster through its representative's Mins / RevMins (src/cluster.cpp:537-539): what travels between GPUs is the
 * representatives' records.  A context owns one communicator; rank 0 makes the id and the host program ships its 128 bytes
 * to the other ranks however it likes (a file, a socket, MPI, torch.distributed). */
#define IOC_DIST_ID_BYTES 128
int ioc_dist_unique_id(uint8_t* id /* [IOC_DIST_ID_BYTES] */);
int ioc_dist_init(ioc_ctx* ctx, const uint8_t* id, int32_t rank, int32_t world);   /* ncclCommInitRank on the context's device */
int ioc_dist_shutdown(ioc_ctx* ctx);
int ioc_dist_info(const ioc_ctx* ctx, int32_t* rank, int32_t* world);              /* (0, 1) without a communicator */
/* collectives on the context's stream.  *_device: device buffers, HBM to HBM; the others stage host data through HBM and
 * return when the result is in the caller's arrays. */
int ioc_dist_allgather_device(ioc_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank);
/* ragged: rank r contributes counts[r] elements of esize bytes, landing at d_recv + displs[r] * esize on every rank; a
 * rank's d_send may be its own slot of d_recv.  counts / displs [world]: host arrays, identical on every rank. */
int ioc_dist_allgatherv_device(ioc_ctx* ctx, const void* d_send, void* d_recv, const int64_t* counts, const int64_t* displs,
                               int32_t esize);
int ioc_dist_allgather_i64(ioc_ctx* ctx, int64_t mine, int64_t* all /* [world] */);
/* ragged host records in rank order; sizes [world] receives the byte counts (recv == NULL: only the sizes) */
int ioc_dist_allgatherv_host(ioc_ctx* ctx, const void* send, int64_t bytes, void* recv, int64_t* sizes);
int ioc_dist_allreduce_max(ioc_ctx* ctx, double* x);
int ioc_dist_barrier(ioc_ctx* ctx);
typedef struct {
    float ms_exchange_lists;   /* the four ragged all-gathers of the minimizer lists (HIP events on the stream) */
    double ms_merge;           /* the one-pass merge on this rank (wall) */
    int64_t bytes_lists;       /* this rank's contribution: minimizer lists ... */
    int64_t bytes_records;     /* ... and per-representative records + raw sequences */
    int32_t sharded;           /* 1: score + resolve were sharded over the ranks (ioc_set_shard) */
    int32_t exchanges;         /* all-reduces the sharded resolve issued */
} ioc_dist_merge_times;
/* Sharded score + resolve (fast mode): with world > 1 the context scores and decides only the queries j with
 * j % world == rank (the same queries, index and forced decisions on every rank), and after each sweep of the resolve the
 * ranks' shares of `valid` and the sweep's control words are combined through `fn`: an in-place all-reduce over `count`
 * elements of DEVICE memory at d_buf, ordered after the work already on the context's stream (hip_stream) — kind
 * IOC_XCHG_MAX_U8 (bytes, maximum), IOC_XCHG_MIN_U32 (words, minimum) or IOC_XCHG_SUM_I32 (words, sum); non-zero = failure.
 * The decisions end up complete on every rank; they are those of the unsharded resolve (a query's decision depends on its
 * own candidates and on `valid` of earlier queries only).  sahlin / furious contexts and ioc_cluster_consensus ignore the
 * setting (their alignment rounds / windowed passes read per-query state of all queries); ioc_scored_candidates of a query
 * another rank owns reports no candidates.  world <= 1 or fn == NULL switches it off.  ioc_dist_merge installs an RCCL
 * exchange by itself (IOC_DIST_SHARD=0 keeps the replicated merge); the hook exists for other transports and for tests. */
#define IOC_XCHG_MAX_U8 0
#define IOC_XCHG_MIN_U32 1
#define IOC_XCHG_SUM_I32 2
typedef int (*ioc_exchange_fn)(void* user, void* d_buf, int64_t count, int32_t kind, void* hip_stream);
int ioc_set_shard(ioc_ctx* ctx, int32_t world, int32_t rank, ioc_exchange_fn fn, void* user);
/* all-reduces issued by the last ioc_resolve (0: it was not sharded) */
int32_t ioc_shard_exchanges(const ioc_ctx* ctx);
/* sahlin / furious: scoring and resolve stay replicated under ioc_set_shard (their tie sets read every query's candidates), the
 * ALIGNMENT rounds are shared out by owner of the query (a pair's result is a function of its two sequences, cluster.cpp:408-459);
 * the verdicts of a round travel as one summed word array through the same hook.  Pairs this rank has aligned since
 * ioc_set_shard: */
int64_t ioc_shard_aligned_pairs(const ioc_ctx* ctx);
/* The RCCL form of that exchange over the context's communicator (ioc_dist_init), on the context's stream: directly, and
 * installed as the context's shard setting with the communicator's world and rank (on = 0 removes it). */
int ioc_dist_exchange(ioc_ctx* ctx, void* d_buf, int64_t count, int32_t kind);
int ioc_dist_set_shard(ioc_ctx* ctx, int32_t on);

/* The merge of ALL ranks' freshly clustered batches (Depth 0), on every rank: the left fold ((b0 + b1) + b2) ... in rank order
 * makes the decisions of ONE greedy loop over the representatives of b0, b1, ... in which b0's are clusters from the start
 * (ioc_batch_view::is_cluster: src/cluster.cpp:178-217 only appends).  reps = THIS rank's cluster representatives, one record
 * each in cluster order (compact lists; min_val / min_pos host or device per reps->minimizers_on_device — e.g. the output of
 * ioc_gather_records_device; raw_seq / raw_off for sahlin / furious).  out_counts [world] receives the clusters of every
 * rank; out_cls / out_strand [sum of counts] the decision for every gathered representative in rank order, as
 * ioc_cluster_merge reports them (call with out_cls == NULL to learn the counts first).  ioc_index_export afterwards returns
 * the merged MinDB. */
int ioc_dist_merge(ioc_ctx* ctx, const ioc_params* p, const char* table_path, const ioc_batch_view* reps, int32_t min_cls_size,
                   int64_t out_cap, int32_t* out_cls, int8_t* out_strand, int64_t* out_counts, ioc_cluster_stats* stats,
                   ioc_dist_merge_times* times);

#ifdef __cplusplus
}
#endif
#endif /* ISONCLUST2_HIP_H */
