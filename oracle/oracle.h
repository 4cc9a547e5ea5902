/*
 * oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * The oracle is a from-scratch CPU restatement of isONclust2's read->cluster assignment
 * path (reference files cited per function in oracle.cpp).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; nothing under
 * isonclust2_amd/ links, imports or calls it.
 *
 * Parity pin: the restatement is checked against every known-answer vector the
 * reference's own unit tests hold for this path (test/isONclust2_test.cpp:17-135,
 * 184-203; tests/golden/reference_kat.json) and, function by function, against the
 * three reference translation units that compile stand-alone (oracle/_ref, see Makefile).
 */
#ifndef IOC_ORACLE_H
#define IOC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t k;               /* CmdArgs::KmerSize        args.h:13 */
    int32_t w;               /* CmdArgs::WindowSize      args.h:16 */
    int32_t min_shared;      /* CmdArgs::MinShared       args.h:17 */
    int32_t min_cls_size;    /* CmdArgs::MinClsSize      args.h:21 */
    int32_t mode;            /* ClsMode: 0 sahlin, 1 fast, 2 furious, 3 none  args.h:7 */
    int32_t cons_max_size;   /* CmdArgs::ConsMaxSize (<=0: consensus off)  args.h:19 */
    double min_qual;         /* CmdArgs::MinQual         args.h:22 */
    double mapped_threshold; /* CmdArgs::MappedThreshold args.h:23 */
    double aligned_threshold;/* CmdArgs::AlignedThreshold args.h:24 */
    double min_fraction;     /* CmdArgs::MinFraction     args.h:25 */
    double min_prob_no_hits; /* CmdArgs::MinProbNoHits   args.h:26 */
} orc_params;

typedef struct {
    uint64_t probes;         /* MinDB.find calls in GetMinimizerHits  (M) */
    uint64_t postings;       /* raw hits emitted                      (H) */
    uint64_t mapped_calls;   /* getMappedRatio calls                  (C_s) */
    uint64_t queries;        /* getBestCluster calls */
    uint64_t new_clusters;
    uint64_t joins;
    uint64_t index_appends;  /* postings appended by AddMinimizers */
    uint64_t aln_invoked;
    uint64_t tie_reads;      /* queries with >=2 passing candidates tied at the winning Size */
    uint64_t cons_invoked;   /* CONS_INVOKED, cluster.cpp:22,295 */
} orc_stats;

/* The partial-order graphs of the consensus (spoa, absent from the reference tree) behind the operations the
 * reference performs on them.  side 0 = leftBatch->ConsGs[idx] (idx = left cluster id), side 1 =
 * rightBatch->ConsGs[idx] (idx = right entry).  size: sequences().size(), < 0 if there is no such graph. */
typedef struct {
    void* user;
    int (*create)(void* user, int side, int idx, const char* seq, int len);                 /* new graph + AddSeqToGraph(seq, 1) */
    int (*size)(void* user, int side, int idx);
    int (*add)(void* user, int side, int idx, const char* seq, int len, unsigned weight);   /* AddSeqToGraph */
    int (*consensus)(void* user, int side, int idx, char* out, int cap);                    /* GenerateConsensus -> length */
    int (*purge)(void* user, int side, int idx, const char* seq, int len, unsigned weight); /* ConsPurge */
} orc_cons_ops;
/* ops == NULL switches the hook off; cons_min_size / cons_period = CmdArgs::ConsMinSize / ConsPeriod */
void orc_set_consensus(const orc_cons_ops* ops, int cons_min_size, int cons_period);

/* ---- alignment fallback (src/cluster.cpp:408-515) --------------------------------------------------
 * The oracle's own scalar semi-global aligner with traceback (parasail itself is absent from the reference tree:
 * tie-breaking parity unpinned beyond AlnRatioTest).  It is what sahlin / furious mode use unless another aligner
 * is put behind the seam with orc_set_aligner (signature: int fn(read, nread, rep, nrep, gap_open, gap_extend,
 * comp_out, comp_cap) -> length of comp or < 0).  orc_use_builtin_aligner(0) restores "status -2 without a hook". */
void orc_set_aligner(void* fn);
void orc_use_builtin_aligner(int on);
int orc_align(const char* read, int nread, const char* rep, int nrep, int match, int mismatch, int gap_open, int gap_extend,
              char* comp, int comp_cap, int* score);
int orc_gap_open(double e);                                                              /* setGapOpen, cluster.cpp:425-440 */
double orc_aln_ratio(const char* comp, int n, double e, unsigned slen, unsigned k);      /* getAlnRatio, cluster.cpp:442-459 */

/* ---- tracing of the intermediate tables (SURVEY §8(c) golden item 3) ----------------------------------
 * orc_trace_set: the right-batch entries whose candidate table is recorded when the greedy loop reaches them
 * (n = 0: none); mapped_calls != 0 also logs every getMappedRatio call of the run.  Both logs are cleared here and
 * filled by the next orc_cluster.  orc_trace_rows: one row per (cls, strand) of the traced entries' hit maps
 * (minimizer.cpp:44-76): Size, Index of the first hit, totalMapped of getMappedRatio (computed for every candidate
 * of a traced entry), position in the SortMinimizerHits order (cluster.cpp:622-636), walked = the reference's walk
 * called getMappedRatio on it.  Call with NULLs to size. */
void orc_trace_set(const int32_t* entries, int n, int mapped_calls);
int64_t orc_trace_rows(int32_t* entry, int32_t* cls, int32_t* strand, uint32_t* size, uint32_t* first_index, uint32_t* total_mapped,
                       int32_t* order_pos, uint8_t* walked);
int64_t orc_trace_mapped_calls(int32_t* entry, int32_t* cls, int32_t* strand, uint32_t* total, uint32_t* hpc_len, double* ratio,
                               double* p_error);

/* ---- primitives ---------------------------------------------------------------- */
int orc_hpc(const char* seq, const char* qual, int n, char* oseq, char* oqual);
int orc_revcomp(const char* seq, int n, char* out);
int orc_kmer_encode(const char* seq, int n, int k, uint32_t* out);
int orc_minimizers(const uint32_t* kmers, int n, int k, int w, uint32_t* omin, uint32_t* opos,
                   uint32_t* oidx);
void orc_qual_tab(int nomin, double* out129);
double orc_qual_score(const char* qual, int n, int k);
double orc_error_rate(const char* qual, int n, int nomin);
double orc_round(double x, int precision);
int orc_pmin_table(const char* binpath, int k, int w, double* out225);
double orc_pmin_lookup(const double* tab225, double e1, double e2, int* err);
int orc_gap_limit(double p_shared, double min_prob_no_hits);
uint32_t orc_kmer_to_index(const char* kmer, int k);
void orc_index_to_kmer(uint32_t idx, int k, char* out);

/* One reference->query scoring exactly as the reference's MinMatchTest drives it
 * (test/isONclust2_test.cpp:85-135). Returns 0 on success. */
int orc_minmatch(const char* ref, const char* refq, int nref, const char* read, const char* readq,
                 int nread, int k, int w, const char* binpath, double min_prob_no_hits,
                 uint32_t* top_size, double* p_error, double* mapped_ratio);

/* ---- read set: FillQualScores + SortByQualScores -------------------------------- */
void* orc_reads_new(const char* seqs, const char* quals, const int64_t* offs, int n);
void orc_reads_free(void* h);
void orc_reads_score_sort(void* h, int k, int w);
int orc_reads_n(void* h);
void orc_reads_shift_orig(void* h, int base); /* test plumbing: original-read ids += base */
void orc_reads_order(void* h, int32_t* orig_index, double* score, double* err);

/* ---- batch: PrepareSortedBatch + ClusterSortedReads ------------------------------ */
void* orc_batch_prepare(void* reads, int start, int end, const orc_params* p, int batch_nr);
void orc_batch_free(void* b);
int orc_batch_n_entries(void* b);
/* per entry (rep of the entry): state 0 = clusterable, 1 = null placeholder (RawSeq == nullptr),
 * orig index of the read, raw/hpc lengths, raw score/error, hpc error, minimizer counts. */
void orc_batch_entry_info(void* b, int32_t* state, int32_t* orig, int32_t* raw_len, int32_t* hpc_len,
                          double* score, double* raw_err, double* hpc_err, int32_t* n_fwd,
                          int32_t* n_rev);
/* copy minimizers of entry i's representative; strand 0 = Mins, 1 = RevMins. returns count */
int orc_batch_entry_mins(void* b, int i, int strand, uint32_t* omin, uint32_t* opos, uint32_t* oidx);
int orc_batch_entry_hpc(void* b, int i, char* oseq, char* oqual);

/* mainCluster + ClusterSortedReads: right == NULL -> single-batch (pseudo batch) clustering.
 * Returns 0, or <0 on the reference's exit(1) conditions. `left` is mutated (and owns the moved
 * members afterwards, like the reference). */
int orc_cluster(void* left, void* right, const orc_params* p, const char* binpath, orc_stats* st);

int orc_batch_n_clusters(void* b);
int orc_batch_n_members(void* b);
/* dump cluster membership as dumpClusters would see it (main.cpp:430-453): for every cluster c
 * and every member m of it (element 0 included): cls[], orig read index[], MatchStrand[],
 * is_rep_copy[] (1 for the synthetic rep_<batch>_<id> element). returns number written. */
int orc_batch_members(void* b, int32_t* cls, int32_t* orig, int32_t* strand, int32_t* is_rep);
/* export the inverted index: returns number of keys; call with NULLs to size. */
int64_t orc_batch_index(void* b, uint32_t* keys, int64_t* offs, uint32_t* postings,
                        int64_t* n_postings);

/* UpdateMinDB (src/minimizer.cpp:124-160) on the batch's index for cluster `cls`. */
int orc_batch_update_mindb(void* b, int cls, const uint32_t* old_min, int64_t n_old, const uint32_t* new_min,
                           int64_t n_new, int mins_too);

#ifdef __cplusplus
}
#endif
#endif
