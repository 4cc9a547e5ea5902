// ioc_host.cpp — host side of the path: the empirical-probability table, the scalar
// thresholds the device compares against, and the ClusterSortedReads driver on flat arrays.
//
// Mirrors (does not copy) the reference's host logic:
//   InitMinSharedMap / GetPMinShared     src/p_emp_prob.cpp:22-94
//   round(number, 2)                     src/util.cpp:6-10
//   gates + new-cluster / join logic     src/cluster.cpp:115-261
//   candidate order of equal Size        src/minimizer.cpp:44-76 + src/cluster.cpp:609-636
//     (libstdc++ unordered_map iteration order + std::sort) — replayed here with the very same
//     container types for the rare queries whose winner depends on it.
// All arithmetic of the hot path itself (hits, Sizes, mapped totals, candidate walk) runs on the
// GPU; there is no CPU fallback: without a device every entry point fails.
#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <utility>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <vector>

#include "ioc_internal.h"

// ---- the worker pool behind ioc_parallel_for ---------------------------------------------------------------------------
namespace {
struct IocPool {
    std::mutex mu, region;
    std::condition_variable cv_work, cv_done;
    const std::function<void(size_t)>* fn = nullptr;
    std::atomic<size_t> next{0};
    size_t count = 0, want = 0, active = 0, n_workers = 0;
    uint64_t gen = 0;
};
IocPool* g_pool = nullptr;  // (never destroyed: its threads wait on it until the process ends)
std::once_flag g_pool_once;
thread_local bool t_pool_worker = false;

void pool_worker(IocPool* P, size_t idx)
{
    t_pool_worker = true;
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(P->mu);
    for (;;) {
        P->cv_work.wait(lk, [&] { return P->gen != seen; });
        seen = P->gen;
        if (idx >= P->want) continue;  // (a region for fewer threads)
        const std::function<void(size_t)>* f = P->fn;
        const size_t cnt = P->count;
        lk.unlock();
        for (size_t x = P->next.fetch_add(1); x < cnt; x = P->next.fetch_add(1)) (*f)(x);
        lk.lock();
        if (--P->active == 0) P->cv_done.notify_one();
    }
}
void spawn_run(size_t count, size_t nt, const std::function<void(size_t)>& f)
{
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            for (size_t x = next.fetch_add(1); x < count; x = next.fetch_add(1)) f(x);
        });
    for (auto& t : th) t.join();
}
}  // namespace

void ioc_pool_run(size_t count, size_t nt, const std::function<void(size_t)>& f)
{
    std::call_once(g_pool_once, [] {
        IocPool* P = new IocPool;
        P->n_workers = std::max<size_t>(1, std::min<size_t>(16, std::thread::hardware_concurrency())) - 1;
        for (size_t i = 0; i < P->n_workers; ++i) std::thread(pool_worker, P, i).detach();
        // (a forked child has no workers: it starts threads per call, as does every region while g_pool is null)
        pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; });
        g_pool = P;
    });
    IocPool* P = g_pool;
    std::unique_lock<std::mutex> region;
    if (P && !t_pool_worker) region = std::unique_lock<std::mutex>(P->region, std::try_to_lock);
    if (!P || !region.owns_lock() || P->n_workers == 0) {
        spawn_run(count, nt, f);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->fn = &f;
        P->count = count;
        P->next.store(0);
        P->want = std::min(nt - 1, P->n_workers);  // the caller works as well
        P->active = P->want;
        P->gen++;
    }
    P->cv_work.notify_all();
    for (size_t x = P->next.fetch_add(1); x < count; x = P->next.fetch_add(1)) f(x);
    std::unique_lock<std::mutex> lk(P->mu);
    P->cv_done.wait(lk, [&] { return P->active == 0; });
    P->fn = nullptr;
}

namespace {

struct PTable {
    double p[225];
    int filled = 0;
};

int load_table(const char* path, int K, int W, PTable& t)
{
    for (auto& x : t.p) x = std::nan("");
    t.filled = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return IOC_ERR_TABLE;
    char magic[8];
    uint32_t n = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "IOCPMIN1", 8) == 0 && fread(&n, 4, 1, f) == 1;
    for (uint32_t c = 0; ok && c < n; ++c) {
        int32_t kw[2];
        double cells[225];
        ok = fread(kw, 4, 2, f) == 2 && fread(cells, 8, 225, f) == 225;
        // rows with k == K and |w_row - W| <= 2; later rows overwrite (p_emp_prob.cpp:37-43)
        if (ok && kw[0] == K && std::abs(kw[1] - W) <= 2)
            for (int i = 0; i < 225; ++i)
                if (!std::isnan(cells[i])) t.p[i] = cells[i];
    }
    fclose(f);
    if (!ok) return IOC_ERR_TABLE;
    for (auto x : t.p)
        if (!std::isnan(x)) t.filled++;
    return IOC_OK;
}

// ---- candidate order replay (only for order-dependent ties) ---------------------------------------
typedef std::pair<int, int> StrandedCluster;
struct StrandedClsHash {  // src/minimizer.h:52-58
    std::size_t operator()(const StrandedCluster& u) const { return size_t(int(u.first * u.second)); }
};
struct HitInfo {
    unsigned Size = 0;
    unsigned Mapped = 0;
};
struct SortedHit {  // src/minimizer.h:84-91
    unsigned Size, Cls;
    int Strand;
};
bool by_size_desc(const std::unique_ptr<SortedHit>& a, const std::unique_ptr<SortedHit>& b)
{
    return a->Size > b->Size;
}

// IOC_TRACE=1: wall-clock of the driver's phases on stderr (developer aid)
struct PhaseTrace {
    bool on = getenv("IOC_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    int64_t calls = 0;  // what the phase did that often (the per-query candidate tables of a round)
    void mark(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        if (calls)
            fprintf(stderr, "[ioc] %-28s %9.3f ms (%lld candidate tables)\n", what, std::chrono::duration<double, std::milli>(now - t).count(),
                    (long long)calls);
        else
            fprintf(stderr, "[ioc] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        calls = 0;
        t = now;
    }
};


struct Cand {
    int32_t cls;  // final cluster id
    int8_t strand;
    uint32_t size, first, mapped;
    int32_t target;
};

}  // namespace

extern "C" {

int ioc_host_gap_limits(const char* table_path, int32_t k, int32_t w, double min_prob_no_hits, int32_t* gap_limit,
                        double* p_shared)
{
    if (!table_path || !gap_limit) return IOC_ERR_ARG;
    PTable t;
    int r = load_table(table_path, k, w, t);
    if (r != IOC_OK) return r;
    if (t.filled != 225) return IOC_ERR_TABLE;  // GetPMinShared would throw (p_emp_prob.cpp:87-89)
    for (int i = 0; i < 225; ++i) {
        const double pe = 1.0 - t.p[i];
        if (p_shared) p_shared[i] = t.p[i];
        // predicate of cluster.cpp:333-347: pow(pError, double(n)) >= MinProbNoHits, monotone in n
        int32_t lim = -1;
        if (pow(pe, 0.0) >= min_prob_no_hits) {
            lim = 0;
            while (lim < INT32_MAX - 2 && pow(pe, double(lim + 1)) >= min_prob_no_hits) {
                lim++;
                if (lim > (1 << 22)) {  // pError == 1 (or p0 <= 0): every gap passes
                    lim = INT32_MAX - 2;
                    break;
                }
            }
        }
        gap_limit[i] = lim;
    }
    return IOC_OK;
}

uint8_t ioc_host_err_cell(double e)
{
    // round(e, 2) = std::round(e * int(pow(10,2))) / 100, then clamp to [0.01, 0.15]; the map keys
    // are c/100 for c = 1..15, so the cell is the clamped integer numerator.
    int decimals = int(std::pow(10, 2));
    double m = std::round(e * decimals);
    if (std::isnan(m)) return 0;
    double r = m / decimals;
    if (r > 0.15) return 15;
    if (r < 0.01) return 1;
    for (int c = 1; c <= 15; ++c)
        if (double(c) / 100 == r) return uint8_t(c);
    return 0;
}

uint32_t ioc_host_min_total(uint32_t hpc_len, double thr)
{
    if (hpc_len == 0) return 0xFFFFFFFEu;
    auto pass = [&](uint32_t T) {
        float mr = float(double(T) / double(hpc_len));  // cluster.cpp:390-399 narrows to float
        return mr >= thr;
    };
    if (!pass(hpc_len)) return 0xFFFFFFFEu;
    uint32_t lo = 0, hi = hpc_len;  // pass(hi) true
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (pass(mid))
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// ---- candidate order replay ---------------------------------------------------------------------------
// Rebuilds the reference's `hitOrder` (SortMinimizerHits over the GetMinimizerHits map) for one query
// from the device-computed hit table: same container types, same insertion order, same std::sort.
struct Ordered {
    int32_t cls;      // final cluster id
    int32_t target;   // device target id
    int8_t strand;
    uint32_t size, mapped;
};

// the device's hit table of one query (IocCandTable, ioc_internal.h), kept so that the host part can run on a thread
using RawCands = IocCandTable;

static int fetch_cands(ioc_ctx* c, int q, RawCands& rc)
{
    std::vector<RawCands> one;
    int r = ioc_query_candidates_many(c, std::vector<int>{q}, one);
    if (r != IOC_OK) return r;
    rc = std::move(one[0]);
    return IOC_OK;
}

// pure host code (reads the context's host arrays only): safe on worker threads
static void order_from(const ioc_ctx* c, const RawCands& rc, const std::vector<int32_t>& cid, std::vector<Ordered>& out)
{
    out.clear();
    const int q = rc.q, nc = int(rc.tg.size());
    const std::vector<int32_t>& tg = rc.tg;
    const std::vector<int8_t>& st = rc.st;
    const std::vector<uint32_t>&sz = rc.sz, &fi = rc.fi, &tm = rc.tm;
    std::vector<Cand> cs;
    cs.reserve(size_t(nc));
    for (int i = 0; i < nc; ++i) {
        int32_t id = tg[size_t(i)] < c->L ? tg[size_t(i)] : cid[size_t(tg[size_t(i)] - c->L)];
        if (id < 0) continue;
        cs.push_back(Cand{id, st[size_t(i)], sz[size_t(i)], fi[size_t(i)], tm[size_t(i)], tg[size_t(i)]});
    }
    // insertion order of the reference: +1 strand raw hits first, in read-minimizer order, posting
    // lists ascending in cluster id (minimizer.cpp:52-73) => by (first hitting Index, cluster id)
    std::stable_sort(cs.begin(), cs.end(), [](const Cand& a, const Cand& b) {
        if (a.strand != b.strand) return a.strand > b.strand;
        if (a.first != b.first) return a.first < b.first;
        return a.cls < b.cls;
    });
    const size_t nm = size_t(c->h_off_fwd[size_t(q) + 1] - c->h_off_fwd[size_t(q)]) +
                      size_t(c->h_off_rev[size_t(q) + 1] - c->h_off_rev[size_t(q)]);
    struct Info {
        unsigned Size = 0, Mapped = 0;
        int32_t Target = -1;
    };
    std::unordered_map<StrandedCluster, Info, StrandedClsHash> res(20 * nm, StrandedClsHash());
    for (auto& x : cs) {
        Info& h = res[std::make_pair(int(x.cls), int(x.strand))];
        h.Size = x.size;
        h.Mapped = x.mapped;
        h.Target = x.target;
    }
    if (getenv("IOC_DEBUG_TIES")) {
        fprintf(stderr, "[ioc] order_from: query %d, %zu minimizers, %zu buckets, %zu keys:", q, nm, res.bucket_count(), res.size());
        for (auto& x : cs) fprintf(stderr, " (%d,%d size %u first %u tgt %d)", x.cls, int(x.strand), x.size, x.first, x.target);
        fprintf(stderr, "\n");
    }
    std::vector<std::unique_ptr<SortedHit>> order;
    order.reserve(res.size());
    for (auto& kv : res) {
        auto p = new SortedHit;
        p->Size = kv.second.Size;
        p->Cls = unsigned(kv.first.first);
        p->Strand = kv.first.second;
        order.push_back(std::unique_ptr<SortedHit>(p));
    }
    std::sort(order.begin(), order.end(), by_size_desc);
    out.reserve(order.size());
    for (auto& o : order) {
        const Info& h = res.at(std::make_pair(int(o->Cls), o->Strand));
        out.push_back(Ordered{int32_t(o->Cls), h.Target, int8_t(o->Strand), o->Size, h.Mapped});
    }
}

static int build_order(ioc_ctx* c, int q, const std::vector<int32_t>& cid, std::vector<Ordered>& out)
{
    RawCands rc;
    int r = fetch_cands(c, q, rc);
    if (r != IOC_OK) return r;
    order_from(c, rc, cid, out);
    return IOC_OK;
}


// first passing candidate in the reference's order (cluster.cpp:381-403) — used for the queries whose
// winner is order-dependent
static int replay_order(ioc_ctx* c, int q, const std::vector<int32_t>& cid, uint32_t need, int32_t& out_cls,
                        int8_t& out_strand, std::vector<std::pair<int32_t, int8_t>>* dep = nullptr)
{
    std::vector<Ordered> order;
    int r = build_order(c, q, cid, order);
    if (r != IOC_OK) return r;
    out_cls = -1;
    out_strand = 0;
    if (getenv("IOC_DEBUG_TIES")) {
        fprintf(stderr, "[ioc] tie replay of query %d (entry %lld, %d left clusters), need %u:", q,
                (long long)(size_t(q) < c->aln_qid.size() ? c->aln_qid[size_t(q)] : 0), c->L, need);
        for (size_t x = 0; x < order.size() && x < 6; ++x)
            fprintf(stderr, " (%d,%d size %u mapped %u tgt %d)", order[x].cls, int(order[x].strand), order[x].size, order[x].mapped, order[x].target);
        fprintf(stderr, "\n");
    }
    if (order.empty()) return IOC_OK;
    const unsigned top = order[0].size;
    if (top < unsigned(c->params.min_shared)) return IOC_OK;
    for (auto& o : order) {
        if (int(o.size) < int(double(top) * c->params.min_fraction)) break;
        if (o.mapped == 0xFFFFFFFEu) continue;  // rejected by the upper bound of totalMapped (k_gap_bounds): fails
        if (o.mapped == 0xFFFFFFFFu)
            return ioc_fail(c, IOC_ERR_STATE, "tie replay met a candidate the device did not evaluate");
        if (o.mapped >= need) {
            out_cls = o.cls;
            out_strand = o.strand;
            // (the candidates the order chooses among: every passing one of the winner's Size)
            if (dep)
                for (auto& o2 : order)
                    if (o2.size == o.size && o2.mapped != 0xFFFFFFFEu && o2.mapped != 0xFFFFFFFFu && o2.mapped >= need)
                        dep->emplace_back(o2.cls, o2.strand);
            return IOC_OK;
        }
    }
    return IOC_OK;
}

// sequences the alignment fallback needs (raw read / representative sequences, raw error rates)
struct SeqAccess {
    const char* r_seq = nullptr;      // right entries, concatenated
    const int64_t* r_off = nullptr;
    const double* r_err = nullptr;    // RawSeq->ErrorRate()
    const char* l_seq = nullptr;      // left cluster representatives
    const int64_t* l_off = nullptr;
    const double* l_err = nullptr;
};

static void revcomp_inplace(std::string& s)
{
    std::reverse(s.begin(), s.end());
    for (auto& ch : s) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
}

// ---- alignment fallback (getBestClusterAln, cluster.cpp:461-515) --------------------------------------
// The alignment of (query, candidate, strand) is a pure function of the two sequences, so results are
// cached and the pairs of many queries are aligned in one GPU batch (ioc_align_pairs).  IOC_ALIGN_HOST=1 (tests
// only) routes the same pairs through the host aligner instead.
struct AlnDriver {
    ioc_ctx* c = nullptr;
    const SeqAccess* sa = nullptr;
    int n = 0;
    bool host_only = false;
    const int32_t* cuts = nullptr;  // per query: ioc_get_cuts of the current decisions (a scheduling hint for the aligner, ioc_aln_pair::reserved)
    std::unordered_map<uint64_t, double> cache;  // (query, tie key) -> getAlnRatio

    static uint64_t key(int q, uint32_t tie) { return (uint64_t(uint32_t(q)) << 32) | tie; }

    int init()
    {
        // IOC_ALIGN_HOST=1 is a TEST knob (the host aligner is the definition the GPU kernels are checked against):
        // nothing routes to it by itself — a window length the GPU aligner does not take is an error
        host_only = getenv("IOC_ALIGN_HOST") != nullptr;
        if (!host_only && c->params.k > 32)
            return ioc_fail(c, IOC_ERR_CAPACITY, "alignment fallback: window length k above 32 is not supported by the GPU aligner");
        if (host_only) return IOC_OK;
        // resident queries: the pool was uploaded once by ioc_resident_set_sequences
        if (c->res_pool_ready && c->L == 0 && sa->r_seq == c->res_seq.data()) return IOC_OK;
        // pool: right entries 0..n-1, then the left representatives
        const int L = c->L;
        std::vector<int64_t> off(size_t(n) + size_t(L) + 1, 0);
        for (int i = 0; i < n; ++i) off[size_t(i) + 1] = off[size_t(i)] + (sa->r_off[i + 1] - sa->r_off[i]);
        for (int t = 0; t < L; ++t) off[size_t(n + t) + 1] = off[size_t(n + t)] + (sa->l_off[t + 1] - sa->l_off[t]);
        if (L == 0) return ioc_align_set_pool(c, n, sa->r_seq + sa->r_off[0], off.data());  // (no copy: the batch's own buffer)
        std::vector<char> pool(size_t(off.back()) + 1);
        if (n > 0) memcpy(pool.data(), sa->r_seq + sa->r_off[0], size_t(off[size_t(n)]));
        if (L > 0) memcpy(pool.data() + off[size_t(n)], sa->l_seq + sa->l_off[0], size_t(off.back() - off[size_t(n)]));
        return ioc_align_set_pool(c, n + L, pool.data(), off.data());
    }

    double err_of(int32_t target) const { return target < c->L ? sa->l_err[target] : sa->r_err[target - c->L]; }

    int host_pair(int q, uint32_t tie, double& ratio)
    {
        const int32_t target = int32_t(tie >> 1);
        const std::string read(sa->r_seq + sa->r_off[q], size_t(sa->r_off[q + 1] - sa->r_off[q]));
        std::string rep;
        if (target < c->L)
            rep.assign(sa->l_seq + sa->l_off[target], size_t(sa->l_off[target + 1] - sa->l_off[target]));
        else
            rep.assign(sa->r_seq + sa->r_off[target - c->L], size_t(sa->r_off[target - c->L + 1] - sa->r_off[target - c->L]));
        if (tie & 1u) revcomp_inplace(rep);
        const double e = sa->r_err[q] + err_of(target);
        std::vector<char> comp(read.size() + rep.size() + 2);
        const int len = ioc_host_align(read.data(), int32_t(read.size()), rep.data(), int32_t(rep.size()), 2, -2,
                                       ioc_host_gap_open(e), 1, comp.data(), int32_t(comp.size()), nullptr);
        if (len < 0) return ioc_fail(c, len, "host alignment failed (sequence too long for the traceback matrix)");
        ratio = ioc_host_aln_ratio(comp.data(), len, e, uint32_t(read.size()), uint32_t(c->params.k));
        return IOC_OK;
    }

    // Sequence identities across calls (ioc_cluster_consensus re-runs the pipeline over the remaining entries after
    // every consensus event): with c->aln_qid / c->aln_lid set, an alignment result is also kept in the context,
    // keyed by (read identity, representative identity, strand) — the same pair is never aligned twice.
    bool persistent_key(int q, uint32_t tie, std::pair<uint64_t, uint64_t>& k) const
    {
        if (c->aln_qid.size() != size_t(n) || c->aln_lid.size() != size_t(c->L)) return false;
        const int32_t target = int32_t(tie >> 1);
        const uint64_t rid = target < c->L ? c->aln_lid[size_t(target)] : c->aln_qid[size_t(target - c->L)];
        k = std::make_pair(c->aln_qid[size_t(q)], (rid << 1) | (tie & 1u));
        return true;
    }

    // make sure every (query, tie key) of `want` is in the cache
    int ensure(std::vector<std::pair<int, uint32_t>>& want)
    {
        std::sort(want.begin(), want.end());
        want.erase(std::unique(want.begin(), want.end()), want.end());
        std::vector<std::pair<int, uint32_t>> todo;
        for (auto& w : want) {
            if (cache.count(key(w.first, w.second))) continue;
            std::pair<uint64_t, uint64_t> pk;
            if (persistent_key(w.first, w.second, pk)) {
                auto it = c->aln_cache.find(pk);
                if (it != c->aln_cache.end()) {
                    cache[key(w.first, w.second)] = it->second;
                    continue;
                }
            }
            todo.push_back(w);
        }
        if (todo.empty()) return IOC_OK;
        if (host_only) {
            for (auto& w : todo) {
                double ratio = 0;
                int r = host_pair(w.first, w.second, ratio);
                if (r != IOC_OK) return r;
                cache[key(w.first, w.second)] = ratio;
            }
            return IOC_OK;
        }
        // Sharded merge (ioc_set_shard): the alignment rounds of sahlin / furious mode are shared out by owner of the QUERY — a pair's
        // result depends on the two sequences only (cluster.cpp:408-459, 498-507), every rank holds the same decisions and hence
        // the same list of pairs —, rank r aligns the pairs of the queries q with q % world == r, and the verdicts travel as one
        // summed word array (1 = aligned, below the threshold; 2 = at or above it; 0 = not mine).  Scoring and resolve stay
        // replicated in these modes: their tie sets read every query's candidates.
        const bool shard = c->shard_world > 1 && c->shard_fn != nullptr;
        std::vector<size_t> mine;
        for (size_t i = 0; i < todo.size(); ++i)
            if (!shard || todo[i].first % c->shard_world == c->shard_rank) mine.push_back(i);
        std::vector<ioc_aln_pair> pairs(mine.size());
        for (size_t x = 0; x < mine.size(); ++x) {
            const size_t i = mine[x];
            const int32_t target = int32_t(todo[i].second >> 1);
            pairs[x].query = todo[i].first;
            pairs[x].ref = target < c->L ? n + target : target - c->L;
            pairs[x].ref_revcomp = int32_t(todo[i].second & 1u);
            // (the query's cut = int(top Size x MinFraction): the candidates aligned are the ones tied at that top Size — a few dozen
            // shared minimizers for a chance candidate, thousands for a read of the same transcript)
            pairs[x].reserved = (cuts && cuts[todo[i].first] > 0 && cuts[todo[i].first] < INT32_MAX) ? cuts[todo[i].first] : 0;
            pairs[x].e = sa->r_err[todo[i].first] + err_of(target);
        }
        std::vector<double> ratio(mine.size());
        // (the ratios are only ever compared with AlignedThreshold: the tracebacks may stop once that comparison is decided)
        const double saved_thr = c->aln_verdict_thr;
        const char* ev = getenv("IOC_ALIGN_VERDICT");
        if (!(ev && ev[0] == '0')) (void)ioc_align_set_verdict_threshold(c, c->params.aligned_threshold);
        int r = mine.empty() ? IOC_OK : ioc_align_pairs(c, int32_t(pairs.size()), pairs.data(), c->params.k, 2, -2, 1, nullptr, nullptr, ratio.data());
        c->aln_verdict_thr = saved_thr;
        if (shard) {
            // (a rank whose alignment failed still joins the exchange — its peers are on their way into it — with nothing to say;
            // the pairs it owed come back as 0 and every rank fails on them together)
            std::vector<int32_t> words(todo.size(), 0);
            if (r == IOC_OK)
                for (size_t x = 0; x < mine.size(); ++x) words[mine[x]] = ratio[x] >= c->params.aligned_threshold ? 2 : 1;
            const int rx = ioc_shard_sum_host(c, words.data(), int64_t(words.size()));
            if (r != IOC_OK) return r;
            if (rx != IOC_OK) return rx;
            c->shard_aln_pairs += int64_t(mine.size());
            for (size_t i = 0; i < todo.size(); ++i) {
                if (words[i] != 1 && words[i] != 2) return ioc_fail(c, IOC_ERR_STATE, "sharded alignment round: a pair came back from no rank (or from two)");
                // (only the comparison with AlignedThreshold is ever read: the owner's verdict stands in for the ratio)
                cache[key(todo[i].first, todo[i].second)] = words[i] == 2 ? std::max(1.0, c->params.aligned_threshold) : -1.0;
            }
            return IOC_OK;
        }
        if (r != IOC_OK) return r;
        for (size_t i = 0; i < todo.size(); ++i) {
            cache[key(todo[i].first, todo[i].second)] = ratio[i];
            std::pair<uint64_t, uint64_t> pk;
            if (persistent_key(todo[i].first, todo[i].second, pk)) c->aln_cache[pk] = ratio[i];
        }
        return IOC_OK;
    }
};

// candidates of query q tied at the top Size among the current clusters (used when the device's 4
// slots overflowed)
static int fetch_ties(ioc_ctx* c, int q, const std::vector<int32_t>& cid, std::vector<uint32_t>& ties)
{
    const int T = c->L + q;
    std::vector<int32_t> tg(size_t(2) * T + 1);
    std::vector<int8_t> st(size_t(2) * T + 1);
    std::vector<uint32_t> sz(size_t(2) * T + 1), fi(size_t(2) * T + 1), tm(size_t(2) * T + 1);
    const int nc = ioc_query_candidates(c, q, 2 * T, tg.data(), st.data(), sz.data(), fi.data(), tm.data());
    if (nc < 0) return nc;
    uint32_t top = 0;
    ties.clear();
    for (int i = 0; i < nc; ++i) {
        const int32_t id = tg[size_t(i)] < c->L ? tg[size_t(i)] : cid[size_t(tg[size_t(i)] - c->L)];
        if (id < 0) continue;
        if (sz[size_t(i)] > top) {
            top = sz[size_t(i)];
            ties.clear();
        }
        if (sz[size_t(i)] == top) ties.push_back((uint32_t(tg[size_t(i)]) << 1) | (st[size_t(i)] < 0 ? 1u : 0u));
    }
    std::sort(ties.begin(), ties.end());
    return IOC_OK;
}

static int run_pipeline(ioc_ctx* c, const std::vector<uint8_t>& gated, const std::vector<uint32_t>& need,
                        int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats, const SeqAccess* sa,
                        const std::vector<uint8_t>* keep_cluster = nullptr)
{
    const int n = c->n;
    int r;
    PhaseTrace tr;
    if ((r = ioc_index_build(c)) != IOC_OK) return r;
    if ((r = ioc_score(c)) != IOC_OK) return r;
    if ((r = ioc_clear_forced(c)) != IOC_OK) return r;
    if (tr.on) (void)ioc_synchronize(c);
    tr.mark("index build + score");
    // gated entries never become clusters: force them out of the target set
    for (int i = 0; i < n; ++i)
        if (gated[size_t(i)] && (r = ioc_force_decision(c, i, -2, 0)) != IOC_OK) return r;
    // entries that ARE clusters already (the leftmost batch of a one-pass merge): decided, never matched
    if (keep_cluster)
        for (int i = 0; i < n; ++i)
            if ((*keep_cluster)[size_t(i)] && (r = ioc_force_decision(c, i, -1, 0)) != IOC_OK) return r;
    const bool aln_mode = c->params.mode == IOC_MODE_SAHLIN || c->params.mode == IOC_MODE_FURIOUS;
    if (aln_mode && (!sa || !sa->r_seq || !sa->r_off || !sa->r_err || (c->L > 0 && (!sa->l_seq || !sa->l_off || !sa->l_err))))
        return ioc_fail(c, IOC_ERR_ARG, "sahlin/furious need the raw sequences for the alignment fallback (cluster.cpp:461-515)");
    int32_t iters = 0, total_iters = 0;
    std::vector<int32_t> tgt(size_t(n) + 1);
    std::vector<int8_t> str(size_t(n) + 1);
    std::vector<uint8_t> flg(size_t(n) + 1);
    std::vector<int32_t> cid(size_t(n) + 1, -1);
    int64_t aln_invoked = 0;
    int32_t aln_rounds = 0;
    // Alignment fallback (cluster.cpp:553-566) for the queries whose mapping finds nothing although
    // top >= MinShared.  The verdict of query i is a function of the candidates tied at its top Size
    // among the clusters that exist when the loop reaches i (and, if two of them align, of the cluster
    // numbering).  Verdicts are computed for ALL such queries at once under the current decisions, handed
    // to the device as conditional decisions, and the resolve is repeated until every verdict was derived
    // from the very tie set (and order) it is applied to.  The first query whose verdict is stale sees a
    // final prefix, so its new verdict is final: the loop ends after at most (#flagged + 1) rounds, in
    // practice a handful.
    AlnDriver ad;
    std::vector<int32_t> v_t;
    std::vector<int8_t> v_s;
    std::vector<std::vector<uint32_t>> v_ties;
    std::vector<uint8_t> order_dep;
    std::vector<uint32_t> tcount, tkeys;
    std::vector<int32_t> cuts;
    if (aln_mode) {
        ad.c = c;
        ad.sa = sa;
        ad.n = n;
        if ((r = ad.init()) != IOC_OK) return r;
        tr.mark("sequence pool upload");
        v_t.assign(size_t(n) + 1, INT32_MIN);
        v_s.assign(size_t(n) + 1, 0);
        v_ties.assign(size_t(n) + 1, std::vector<uint32_t>());
        order_dep.assign(size_t(n) + 1, 0);
        tcount.assign(size_t(n) + 1, 0);
        tkeys.assign(size_t(n) * IOC_TIE_SLOTS + IOC_TIE_SLOTS, 0);
        if ((r = ioc_set_aln_verdicts(c, v_t.data(), v_s.data())) != IOC_OK) return r;
    }
    for (int round = 0;; ++round) {
        if ((r = ioc_resolve(c, &iters)) != IOC_OK) return r;
        total_iters += iters;
        if ((r = ioc_get_decisions(c, tgt.data(), str.data(), flg.data())) != IOC_OK) return r;
        tr.mark("resolve + decisions");
        if (!aln_mode) break;
        if (round > 2 * n + 8) return ioc_fail(c, IOC_ERR_STATE, "alignment fallback did not converge");
        if ((r = ioc_get_ties(c, tcount.data(), tkeys.data())) != IOC_OK) return r;
        if (!ad.host_only) {
            cuts.resize(size_t(n) + 1);
            if ((r = ioc_get_cuts(c, cuts.data())) != IOC_OK) return r;
            ad.cuts = cuts.data();
        }
        int32_t next_id = c->L;
        for (int i = 0; i < n; ++i) cid[size_t(i)] = (!gated[size_t(i)] && tgt[size_t(i)] < 0) ? next_id++ : -1;
        // queries that reach the alignment and whose verdict is missing, stale or order-dependent
        std::vector<int> bad;
        std::vector<std::vector<uint32_t>> cur;
        std::vector<std::pair<int, uint32_t>> want;
        aln_invoked = 0;
        for (int i = 0; i < n; ++i) {
            if (gated[size_t(i)] || !(flg[size_t(i)] & 2)) continue;
            aln_invoked++;
            std::vector<uint32_t> ties;
            if (tcount[size_t(i)] <= IOC_TIE_SLOTS) {
                ties.assign(tkeys.begin() + size_t(i) * IOC_TIE_SLOTS, tkeys.begin() + size_t(i) * IOC_TIE_SLOTS + tcount[size_t(i)]);
                std::sort(ties.begin(), ties.end());
            } else if ((r = fetch_ties(c, i, cid, ties)) != IOC_OK) {
                return r;
            }
            if (v_t[size_t(i)] != INT32_MIN && ties == v_ties[size_t(i)] && !order_dep[size_t(i)]) continue;
            bad.push_back(i);
            for (uint32_t t : ties) want.emplace_back(i, t);
            cur.push_back(std::move(ties));
            if (ad.host_only && v_t[size_t(i)] == INT32_MIN) break;  // host aligner: no speculation beyond the first
        }
        // Speculation: a tie that is itself a query waiting for its verdict in THIS round stops being a representative if one of its
        // own ties aligns — and the query's tie set then falls back, usually to that very cluster (a read whose closest earlier read
        // is another read of its transcript that is about to join the transcript's cluster).  Such pairs used to make a second
        // alignment round of a few dozen pairs, which costs as much as its longest alignment's critical path (~17 ms for 16.7 kb
        // reads) whatever its size; aligned now, in the batch that runs anyway, they are in the driver's cache when the next
        // resolve asks for them.  Results only ever come out of the cache under the exact (query, tie key) they were aligned for.
        size_t n_spec = 0;
        if (!ad.host_only && !bad.empty() && !(getenv("IOC_ALIGN_SPECULATE") && atoi(getenv("IOC_ALIGN_SPECULATE")) == 0)) {
            std::unordered_map<int, size_t> waiting;
            for (size_t b = 0; b < bad.size(); ++b) waiting[bad[b]] = b;
            const size_t asked = want.size();
            // (only where both steps look like reads of one transcript — their cuts, the top counts of shared minimizers, are not far
            // below the round's median: a chance candidate of a chance candidate is nobody's fallback)
            int64_t med = 0;
            if (ad.cuts) {
                std::vector<int32_t> hp;
                for (int q : bad)
                    if (ad.cuts[q] > 0 && ad.cuts[q] < INT32_MAX) hp.push_back(ad.cuts[q]);
                if (!hp.empty()) {
                    std::nth_element(hp.begin(), hp.begin() + long(hp.size() / 2), hp.end());
                    med = hp[hp.size() / 2];
                }
            }
            auto alike = [&](int q) { return ad.cuts && ad.cuts[q] > 0 && ad.cuts[q] < INT32_MAX && int64_t(ad.cuts[q]) * 4 >= med; };
            for (size_t b = 0; b < bad.size() && n_spec * 4 < asked + 64; ++b)
                for (uint32_t t : cur[b]) {
                    const int32_t target = int32_t(t >> 1);
                    if (target < c->L || !alike(bad[b])) continue;
                    auto it = waiting.find(target - c->L);
                    if (it == waiting.end() || !alike(target - c->L)) continue;
                    for (uint32_t t2 : cur[it->second]) {
                        want.emplace_back(bad[b], (t2 & ~1u) | ((t ^ t2) & 1u));  // (the strands compose)
                        ++n_spec;
                    }
                }
        }
        if (getenv("IOC_DEBUG_ROUND2") && round > 0) {
            for (size_t b = 0; b < bad.size() && b < 12; ++b) {
                const int i = bad[b];
                fprintf(stderr, "[round %d] query %d: verdict so far %d, old ties [", round, i, v_t[size_t(i)]);
                for (uint32_t t : v_ties[size_t(i)]) fprintf(stderr, " %u%s(v %d)", t >> 1, (t & 1) ? "-" : "+", int32_t(t >> 1) >= c->L ? v_t[size_t(int32_t(t >> 1) - c->L)] : -9);
                fprintf(stderr, " ] new ties [");
                for (uint32_t t : cur[b]) fprintf(stderr, " %u%s", t >> 1, (t & 1) ? "-" : "+");
                fprintf(stderr, " ] order_dep %d\n", int(order_dep[size_t(i)]));
            }
        }
        tr.mark("tie sets");
        if (bad.empty()) break;
        aln_rounds++;
        if (getenv("IOC_TRACE") && n_spec) fprintf(stderr, "[ioc] alignment round %d: %zu pairs asked for, %zu more on speculation\n", aln_rounds, want.size() - n_spec, n_spec);
        if ((r = ad.ensure(want)) != IOC_OK) return r;
        tr.mark("alignment batch");
        bool changed = false;
        // several candidates of a query align: the first one in the reference's hitOrder wins (cluster.cpp:481-511).
        // That order is a std::sort over the iteration order of an unordered_map of ALL the query's hits (thousands
        // on a large batch): the hit tables are fetched from the device one after the other, the containers are
        // rebuilt on the host's cores.
        std::vector<std::vector<uint32_t>> passes(bad.size());
        std::vector<size_t> multi;
        for (size_t b = 0; b < bad.size(); ++b) {
            for (uint32_t t : cur[b])
                if (ad.cache[AlnDriver::key(bad[b], t)] >= c->params.aligned_threshold) passes[b].push_back(t);
            if (passes[b].size() > 1) multi.push_back(b);
        }
        std::vector<RawCands> raws;
        {
            std::vector<int> mq(multi.size());
            for (size_t x = 0; x < multi.size(); ++x) mq[x] = bad[multi[x]];
            if ((r = ioc_query_candidates_many(c, mq, raws)) != IOC_OK) return r;  // one launch per chunk of queries
        }
        tr.calls += int64_t(multi.size());
        if (!multi.empty()) tr.mark("  candidate tables fetched");
        tr.calls += int64_t(multi.size());
        std::vector<int32_t> win_t(multi.size(), -1);
        std::vector<int8_t> win_s(multi.size(), 0);
        ioc_parallel_for(multi.size(), [&](size_t x) {
            std::vector<Ordered> order;
            order_from(c, raws[x], cid, order);
            const std::vector<uint32_t>& pass = passes[multi[x]];
            const unsigned top = order.empty() ? 0 : order[0].size;
            for (auto& o : order) {
                if (o.size < top) break;
                const uint32_t k = (uint32_t(o.target) << 1) | (o.strand < 0 ? 1u : 0u);
                if (std::find(pass.begin(), pass.end(), k) != pass.end()) {
                    win_t[x] = o.target;
                    win_s[x] = o.strand;
                    break;
                }
            }
            std::vector<int32_t>().swap(raws[x].tg);  // (the tables are large)
            std::vector<uint32_t>().swap(raws[x].sz);
            std::vector<uint32_t>().swap(raws[x].fi);
            std::vector<uint32_t>().swap(raws[x].tm);
        });
        size_t mx = 0;
        for (size_t b = 0; b < bad.size(); ++b) {
            const int i = bad[b];
            const std::vector<uint32_t>& ties = cur[b];
            const std::vector<uint32_t>& pass = passes[b];
            int32_t vt = -1;
            int8_t vs = 0;
            order_dep[size_t(i)] = pass.size() > 1;
            if (pass.size() == 1) {
                vt = int32_t(pass[0] >> 1);
                vs = (pass[0] & 1u) ? -1 : 1;
            } else if (pass.size() > 1) {
                vt = win_t[mx];
                vs = win_s[mx];
                ++mx;
                if (vt < 0) return ioc_fail(c, IOC_ERR_STATE, "aligned candidate missing from the candidate order");
            }
            if (vt != v_t[size_t(i)] || vs != v_s[size_t(i)] || ties != v_ties[size_t(i)]) changed = true;
            v_t[size_t(i)] = vt;
            v_s[size_t(i)] = vs;
            v_ties[size_t(i)] = ties;
        }
        tr.mark("verdicts");
        if (!changed) break;
        if ((r = ioc_set_aln_verdicts(c, v_t.data(), v_s.data())) != IOC_OK) return r;
    }
    if (aln_mode && (r = ioc_set_aln_verdicts(c, nullptr, nullptr)) != IOC_OK) return r;
    // final cluster ids in creation order (newId = cls.size(), cluster.cpp:178)
    if (c->want_dep_sets) c->last_dep_set.assign(size_t(n), std::vector<std::pair<int32_t, int8_t>>());
    std::fill(cid.begin(), cid.end(), -1);
    int32_t next = c->L;
    for (int i = 0; i < n; ++i)
        if (!gated[size_t(i)] && tgt[size_t(i)] < 0) cid[size_t(i)] = next++;
    int64_t joined = 0, ngated = 0, ties = 0;
    // sharded score + resolve: the candidate table of a query lives on the rank that owns it, so that rank replays the
    // query's order; the winners travel in one all-reduce afterwards (every rank sees the same flags, hence the same slots)
    const bool sharded = c->scored_sharded;
    std::vector<int> tie_q;
    std::vector<int32_t> tie_words;
    for (int i = 0; i < n; ++i) {
        if (gated[size_t(i)]) {
            out_cls[i] = -1;
            out_strand[i] = 0;
            ngated++;
            continue;
        }
        if (tgt[size_t(i)] < 0) {
            out_cls[i] = cid[size_t(i)];
            out_strand[i] = 1;
            continue;
        }
        int32_t t = tgt[size_t(i)];
        int8_t s = str[size_t(i)];
        int32_t cls = t < c->L ? t : cid[size_t(t - c->L)];
        if (getenv("IOC_DEBUG_TIES") && size_t(i) < c->aln_qid.size() && (long long)c->aln_qid[size_t(i)] == atoll(getenv("IOC_DEBUG_TIES")))
        {
            fprintf(stderr, "[ioc] decision of query %d (entry %lld): target %d strand %d flags %u -> cluster %d, need %u;", i, (long long)c->aln_qid[size_t(i)], t, int(s),
                    unsigned(flg[size_t(i)]), cls, need[size_t(i)]);
            RawCands rc;
            if (fetch_cands(c, i, rc) == IOC_OK)
                for (size_t x = 0; x < rc.tg.size(); ++x)
                    if (rc.sz[x] >= 100) fprintf(stderr, " (tgt %d strand %d size %u first %u mapped %u)", rc.tg[x], int(rc.st[x]), rc.sz[x], rc.fi[x], rc.tm[x]);
            fprintf(stderr, "\n");
        }
        if (flg[size_t(i)] & 1) {
            int32_t rt = -1;
            int8_t rs = 0;
            const bool mine = !sharded || (i % c->shard_world) == c->shard_rank;
            if (mine) {
                if ((r = replay_order(c, i, cid, need[size_t(i)], rt, rs, c->want_dep_sets ? &c->last_dep_set[size_t(i)] : nullptr)) != IOC_OK) return r;
                if (rt < 0) return ioc_fail(c, IOC_ERR_STATE, "tie replay found no passing candidate");
            }
            if (sharded) {
                tie_q.push_back(i);
                tie_words.push_back(mine ? rt + 1 : 0);  // (+1: cluster 0 is a value, zero is "not mine")
                tie_words.push_back(mine ? int32_t(rs) : 0);
                rt = 0;
            }
            cls = rt;
            s = rs;
            ties++;
        }
        if (cls < 0) return ioc_fail(c, IOC_ERR_STATE, "joined target is not a cluster");
        out_cls[i] = cls;
        out_strand[i] = s;
        joined++;
    }
    if (sharded && !tie_q.empty()) {
        if ((r = ioc_shard_sum_host(c, tie_words.data(), int64_t(tie_words.size()))) != IOC_OK) return r;
        for (size_t x = 0; x < tie_q.size(); ++x) {
            if (tie_words[2 * x] <= 0) return ioc_fail(c, IOC_ERR_STATE, "tie replay: no rank reported a winner");
            out_cls[tie_q[x]] = tie_words[2 * x] - 1;
            out_strand[tie_q[x]] = int8_t(tie_words[2 * x + 1]);
        }
    }
    tr.mark("final ids + tie replays");
    // (what a caller that re-uses decisions under a changed state — ioc_cluster_consensus — has to know: these decisions
    // depend on which other (cluster, strand) keys exist at all, not only on the candidates they look at)
    if (c->want_dep_sets) c->last_order_dep.assign(size_t(n), 0);
    for (int i = 0; c->want_dep_sets && i < n; ++i) {
        const bool aln_dep = !order_dep.empty() && order_dep[size_t(i)];
        c->last_order_dep[size_t(i)] = uint8_t(((flg[size_t(i)] & 1) ? 1 : 0) | (aln_dep ? 1 : 0));
        if (aln_dep && !(flg[size_t(i)] & 1))  // the tied candidates that align (cluster.cpp:481-511)
            for (uint32_t t : v_ties[size_t(i)])
                if (ad.cache[AlnDriver::key(i, t)] >= c->params.aligned_threshold) {
                    const int32_t tg2 = int32_t(t >> 1);
                    const int32_t id = tg2 < c->L ? tg2 : cid[size_t(tg2 - c->L)];
                    if (id >= 0) c->last_dep_set[size_t(i)].emplace_back(id, int8_t((t & 1u) ? -1 : 1));
                }
    }
    if (stats) {
        stats->n_clusters = next;
        stats->n_joined = joined;
        stats->n_gated = ngated;
        stats->n_tie_replays = ties;
        stats->n_aln_invoked = aln_invoked;
        stats->resolve_iters = total_iters;
        stats->aln_rounds = aln_rounds;
        stats->n_aln_pairs = int64_t(ad.cache.size());
        stats->n_aln_order_dep = 0;
        for (uint8_t x : order_dep) stats->n_aln_order_dep += x;
    }
    return IOC_OK;
}

int ioc_cluster_batch(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_batch_view* rb,
                      int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats)
{
    return ioc_cluster_merge(c, p, table_path, nullptr, rb, out_cls, out_strand, stats);
}

static int cluster_merge_one(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                             const ioc_batch_view* rb, int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats);

// A right batch of more entries than one device pass takes (131 072: the all-pairs candidate tables grow with the square of the
// entries) is the reference's loop all the same (cluster.cpp:115: one entry after the other against whatever clusters exist by
// then): the entries go through in CHUNKS, in order, each against the left state the chunks before it have left behind — the
// MinDB exported after a chunk (ioc_index_export: AddMinimizers of every representative so far) and the representatives'
// error rates and sequences become the next chunk's left batch, exactly what `cluster -l L -r R` does between two processes.
// Cluster ids need no translation: a chunk numbers its new clusters from the count it was given.  IOC_MERGE_CHUNK (entries).
static int cluster_merge_chunked(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_left_view* left, const ioc_batch_view* rb,
                                 int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats, int32_t chunk)
{
    const int n = rb->n;
    const bool seqs = p->mode == IOC_MODE_SAHLIN || p->mode == IOC_MODE_FURIOUS;
    if (left && left->n_keys == -1 && !left->keys) return ioc_fail(c, IOC_ERR_CAPACITY, "a resident left state cannot be combined with a right batch that runs in chunks");
    if (seqs && (!rb->raw_seq || !rb->raw_off)) return ioc_fail(c, IOC_ERR_ARG, "sahlin / furious mode needs the raw sequences");
    // the left state as it grows: owned copies from the second chunk on
    int32_t L = left ? left->n_clusters : 0;
    std::vector<double> l_hpc, l_raw;
    std::vector<uint32_t> keys, post;
    std::vector<int64_t> offs, l_off(1, 0);
    std::string l_seq;
    if (L > 0) {
        l_hpc.assign(left->cls_hpc_err, left->cls_hpc_err + L);
        if (seqs) {
            if (!left->rep_seq || !left->rep_off || !left->cls_raw_err) return ioc_fail(c, IOC_ERR_ARG, "sahlin / furious mode needs the left representatives' sequences");
            l_raw.assign(left->cls_raw_err, left->cls_raw_err + L);
            l_off.assign(left->rep_off, left->rep_off + L + 1);
            l_seq.assign(left->rep_seq + l_off[0], size_t(l_off[size_t(L)] - l_off[0]));
            for (auto& o : l_off) o -= left->rep_off[0];
        }
    }
    ioc_cluster_stats total{};
    ioc_left_view lv{};
    const ioc_left_view* cur = left;
    for (int a = 0; a < n; a += chunk) {
        const int m = std::min<int>(chunk, n - a);
        ioc_batch_view sub = *rb;  // (the minimizer arrays and their absolute offsets stay: a chunk is a window of the per-entry arrays)
        sub.n = m;
        sub.off_fwd = rb->off_fwd + a;
        sub.off_rev = rb->off_rev + a;
        sub.raw_len = rb->raw_len + a;
        sub.hpc_len = rb->hpc_len + a;
        sub.score = rb->score + a;
        sub.raw_err = rb->raw_err + a;
        sub.hpc_err = rb->hpc_err + a;
        sub.state = rb->state ? rb->state + a : nullptr;
        sub.raw_off = rb->raw_off ? rb->raw_off + a : nullptr;
        sub.n_members = rb->n_members ? rb->n_members + a : nullptr;
        sub.is_cluster = rb->is_cluster ? rb->is_cluster + a : nullptr;
        ioc_cluster_stats st{};
        int r = cluster_merge_one(c, p, table_path, cur, &sub, out_cls + a, out_strand + a, &st);
        if (r != IOC_OK) return r;
        total.n_joined += st.n_joined, total.n_gated += st.n_gated, total.n_tie_replays += st.n_tie_replays, total.n_aln_invoked += st.n_aln_invoked;
        total.n_aln_pairs += st.n_aln_pairs, total.n_aln_order_dep += st.n_aln_order_dep, total.aln_rounds += st.aln_rounds;
        total.resolve_iters = std::max(total.resolve_iters, st.resolve_iters);
        total.n_clusters = st.n_clusters;
        if (a + m >= n) break;
        // ---- the next chunk's left batch ----
        for (int i = 0; i < m; ++i)
            if (out_cls[a + i] == L) {  // (opened a cluster: ids are handed out in entry order)
                ++L;
                l_hpc.push_back(rb->hpc_err[a + i]);
                if (seqs) {
                    l_raw.push_back(rb->raw_err[a + i]);
                    l_seq.append(rb->raw_seq + rb->raw_off[a + i], size_t(rb->raw_off[a + i + 1] - rb->raw_off[a + i]));
                    l_off.push_back(int64_t(l_seq.size()));
                }
            }
        if (int64_t(L) != st.n_clusters) return ioc_fail(c, IOC_ERR_STATE, "chunked clustering: cluster count out of step");
        int64_t nk = 0, np = 0;
        if ((r = ioc_index_export(c, &nk, &np, nullptr, nullptr, nullptr)) != IOC_OK) return r;
        keys.assign(size_t(nk) + 1, 0);
        post.assign(size_t(np) + 1, 0);
        offs.assign(size_t(nk) + 2, 0);
        if ((r = ioc_index_export(c, &nk, &np, keys.data(), offs.data(), post.data())) != IOC_OK) return r;
        lv = ioc_left_view{};
        lv.n_clusters = L;
        lv.cls_hpc_err = l_hpc.data();
        lv.n_keys = nk;
        lv.keys = keys.data();
        lv.offs = offs.data();
        lv.postings = post.data();
        lv.rep_seq = seqs ? l_seq.data() : nullptr;
        lv.rep_off = seqs ? l_off.data() : nullptr;
        lv.cls_raw_err = seqs ? l_raw.data() : nullptr;
        cur = L > 0 ? &lv : nullptr;
    }
    if (stats) *stats = total;
    return IOC_OK;
}

int ioc_cluster_merge(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                      const ioc_batch_view* rb, int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats)
{
    if (!c || !p || !table_path || !rb || !out_cls || !out_strand) return IOC_ERR_ARG;
    int32_t chunk = 131072;
    if (const char* e = getenv("IOC_MERGE_CHUNK")) chunk = std::max(1, std::min(131072, atoi(e)));
    c->chunked_call = false;
    if (rb->n > chunk && p->mode != IOC_MODE_NONE) {
        const int r = cluster_merge_chunked(c, p, table_path, left, rb, out_cls, out_strand, stats, chunk);
        c->chunked_call = true;  // (the context's queries are the LAST chunk's: ioc_gather_records_device says so)
        return r;
    }
    return cluster_merge_one(c, p, table_path, left, rb, out_cls, out_strand, stats);
}

static int cluster_merge_one(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                             const ioc_batch_view* rb, int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats)
{
    const int32_t L = left ? left->n_clusters : 0;
    if (L < 0) return ioc_fail(c, IOC_ERR_ARG, "negative left cluster count");
    const int n = rb->n;
    if (n < 0) return ioc_fail(c, IOC_ERR_ARG, "negative batch size");
    PhaseTrace tr0;
    int32_t glim[225];
    int r = IOC_OK;
    {   // (the consensus driver comes here once per pass with the same table: the file and its 225 pow() searches are kept)
        static std::mutex mu;
        static std::string key;
        static int32_t kept[225];
        char tag[96];
        snprintf(tag, sizeof tag, "|%d|%d|%.17g", p->k, p->w, p->min_prob_no_hits);
        const std::string want = std::string(table_path) + tag;
        std::lock_guard<std::mutex> lk(mu);
        if (key != want) {
            key.clear();
            r = ioc_host_gap_limits(table_path, p->k, p->w, p->min_prob_no_hits, kept, nullptr);
            if (r == IOC_OK) key = want;
        }
        if (r == IOC_OK) memcpy(glim, kept, sizeof glim);
    }
    if (r != IOC_OK) return ioc_fail(c, r, "empirical probability lookup failure (k, w outside the table)");
    if ((r = ioc_set_params(c, p, glim)) != IOC_OK) return r;
    if (p->mode == IOC_MODE_NONE) {
        // no branch of getBestCluster fires for None (cluster.cpp:545-567): every clusterable entry
        // opens its own cluster.  Same gates, no device work needed — but keep the contract that
        // nothing runs without a device: the context exists, so a device is present.
    }
    // ---- gates of the loop, in the reference's order (cluster.cpp:116-160) ----
    std::vector<uint8_t> gated(size_t(n) + 1, 0), cell(size_t(n) + 1, 1);
    std::vector<uint32_t> need(size_t(n) + 1, 0xFFFFFFFEu);
    bool compact = false;
    for (int i = 0; i < n; ++i) {
        bool g = false;
        if (rb->n_members && rb->depth > 0 && rb->min_cls_size > 1 && rb->n_members[i] < rb->min_cls_size)
            g = true;                                                               // :119-123
        else if (rb->state && rb->state[i] == 1) g = true;                          // null rep
        else if (rb->state && rb->state[i] == 2)
            return ioc_fail(c, IOC_ERR_INPUT, "entry without HpcSeq (the reference dereferences null here)");
        else if (rb->score[i] < 0) g = true;                                       // :145
        else if (rb->raw_len[i] < uint32_t(2 * p->k)) g = true;                    // :148
        else if (rb->hpc_len[i] < uint32_t(2 * p->k)) g = true;                    // :152
        else if ((-10 * log10(rb->raw_err[i])) <= rb->min_qual) g = true;          // :157
        gated[size_t(i)] = g;
        int64_t nf = rb->off_fwd[i + 1] - rb->off_fwd[i], nr = rb->off_rev[i + 1] - rb->off_rev[i];
        if (g && (nf || nr)) compact = true;
        if (!g) {
            uint8_t cl = ioc_host_err_cell(rb->hpc_err[i]);
            if (cl == 0) return ioc_fail(c, IOC_ERR_TABLE, "error rate is NaN");
            cell[size_t(i)] = cl;
            need[size_t(i)] = ioc_host_min_total(rb->hpc_len[i], p->mapped_threshold);
        }
    }
    if (p->mode == IOC_MODE_NONE) {
        int32_t next = L;
        int64_t ng = 0;
        for (int i = 0; i < n; ++i) {
            out_cls[i] = gated[size_t(i)] ? -1 : next++;
            out_strand[i] = gated[size_t(i)] ? 0 : 1;
            ng += gated[size_t(i)];
        }
        if (stats) *stats = ioc_cluster_stats{next, 0, ng, 0, 0, 0};
        return IOC_OK;
    }
    struct UploadGuard {
        ioc_ctx* c;
        ~UploadGuard() { (void)ioc_wait_uploads(c, 2); }
    } uploads{c};
    if (rb->minimizers_on_device) {
        if (compact) return ioc_fail(c, IOC_ERR_INPUT, "minimizers_on_device: entries skipped by the gates must carry no minimizers");
        r = ioc_queries_upload_devmins(c, n, rb->off_fwd, rb->off_rev, rb->min_val, rb->min_pos, rb->total, rb->hpc_len, cell.data(),
                                       need.data());
    } else if (!compact) {
        // (the caller's arrays stay valid until this function returns: part of the upload may run under the first kernels;
        // `uploads` makes sure it has ended on every way out)
        c->defer_uploads = true;
        r = ioc_queries_upload(c, n, rb->off_fwd, rb->off_rev, rb->min_val, rb->min_pos, rb->total, rb->hpc_len,
                               cell.data(), need.data());
        c->defer_uploads = false;
    } else {
        // slow path: gated entries that still carry minimizers are given empty lists
        std::vector<int64_t> of(size_t(n) + 1, 0), orv(size_t(n) + 1, 0);
        std::vector<uint32_t> mv, mp;
        int64_t tot = 0;
        for (int i = 0; i < n; ++i) {
            of[size_t(i)] = tot;
            if (!gated[size_t(i)]) tot += rb->off_fwd[i + 1] - rb->off_fwd[i];
        }
        of[size_t(n)] = tot;
        for (int i = 0; i < n; ++i) {
            orv[size_t(i)] = tot;
            if (!gated[size_t(i)]) tot += rb->off_rev[i + 1] - rb->off_rev[i];
        }
        orv[size_t(n)] = tot;
        mv.resize(size_t(tot) + 1);
        mp.resize(size_t(tot) + 1);
        for (int i = 0; i < n; ++i) {
            if (gated[size_t(i)]) continue;
            size_t a = size_t(rb->off_fwd[i]), b = size_t(rb->off_fwd[i + 1]);
            std::copy(rb->min_val + a, rb->min_val + b, mv.begin() + of[size_t(i)]);
            std::copy(rb->min_pos + a, rb->min_pos + b, mp.begin() + of[size_t(i)]);
            a = size_t(rb->off_rev[i]);
            b = size_t(rb->off_rev[i + 1]);
            std::copy(rb->min_val + a, rb->min_val + b, mv.begin() + orv[size_t(i)]);
            std::copy(rb->min_pos + a, rb->min_pos + b, mp.begin() + orv[size_t(i)]);
        }
        // the fwd block must precede... offsets are absolute, any layout is valid
        std::vector<int64_t> of2(of), or2(orv);
        // CSR needs off[i+1] as the end of entry i: with gated entries skipped the arrays above
        // already satisfy that (gated entries have zero length).
        r = ioc_queries_upload(c, n, of2.data(), or2.data(), mv.data(), mp.data(), tot, rb->hpc_len, cell.data(),
                               need.data());
    }
    if (r != IOC_OK) return r;
    tr0.mark("gates + query upload");
    if (left && left->n_keys == -1 && !left->keys) {
        // the left state stays as it is on the device (persisted MinDB + any ioc_index_update)
        if (L != c->L) return ioc_fail(c, IOC_ERR_STATE, "resident left state holds a different number of clusters");
        c->built = c->scored = c->resolved = false;
        r = IOC_OK;
    } else if (L > 0) {
        std::vector<uint8_t> lcell(size_t(L), 1);
        for (int i = 0; i < L; ++i) {
            lcell[size_t(i)] = ioc_host_err_cell(left->cls_hpc_err[i]);
            if (lcell[size_t(i)] == 0) return ioc_fail(c, IOC_ERR_TABLE, "left error rate is NaN");
        }
        r = ioc_left_load(c, L, lcell.data(), left->n_keys, left->keys, left->offs, left->postings);
    } else {
        r = ioc_left_load(c, 0, nullptr, 0, nullptr, nullptr, nullptr);
    }
    if (r != IOC_OK) return r;
    tr0.mark("left state load");
    SeqAccess sa;
    sa.r_seq = rb->raw_seq;
    sa.r_off = rb->raw_off;
    sa.r_err = rb->raw_err;
    if (left) {
        sa.l_seq = left->rep_seq;
        sa.l_off = left->rep_off;
        sa.l_err = left->cls_raw_err;
    }
    // furious mode skips the mapping test altogether (cluster.cpp:545-551): no candidate can pass
    if (p->mode == IOC_MODE_FURIOUS) {
        std::vector<uint32_t> never(size_t(n) + 1, 0xFFFFFFFEu);
        if (n > 0) {
            hipError_t e = hipMemcpy(const_cast<uint32_t*>(c->d_min_total), never.data(), size_t(n) * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, hipGetErrorString(e));
        }
    }
    // one-pass merge: the leftmost batch's clusters are not matched, they are clusters (see ioc_batch_view::is_cluster)
    std::vector<uint8_t> keep_cluster;
    if (rb->is_cluster) {
        keep_cluster.assign(rb->is_cluster, rb->is_cluster + n);
        for (int i = 0; i < n; ++i)
            if (keep_cluster[size_t(i)] && gated[size_t(i)]) return ioc_fail(c, IOC_ERR_INPUT, "is_cluster: a gated entry cannot be a cluster");
    }
    return run_pipeline(c, gated, need, out_cls, out_strand, stats, &sa, keep_cluster.empty() ? nullptr : &keep_cluster);
}

int ioc_resident_set_sequences(ioc_ctx* c, const char* raw_seq, const int64_t* raw_off, const double* raw_err)
{
    if (!c || !raw_seq || !raw_off || !raw_err) return IOC_ERR_ARG;
    const int n = c->n;
    if (raw_off[0] != 0) return ioc_fail(c, IOC_ERR_ARG, "raw_off must start at 0");
    for (int i = 0; i < n; ++i)
        if (raw_off[i + 1] < raw_off[i]) return ioc_fail(c, IOC_ERR_ARG, "raw_off must be ascending");
    c->res_seq.assign(raw_seq, size_t(raw_off[n]));
    c->res_off.assign(raw_off, raw_off + n + 1);
    c->res_err.assign(raw_err, raw_err + n);
    c->have_res_seq = true;
    // the alignment fallback's sequence pool stays in HBM with the queries (ioc_align_set_pool clears the flag)
    int r = ioc_align_set_pool(c, n, c->res_seq.data(), c->res_off.data());
    if (r != IOC_OK) return r;
    c->res_pool_ready = getenv("IOC_ALIGN_HOST") == nullptr;
    return IOC_OK;
}

int ioc_cluster_resident(ioc_ctx* c, int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats)
{
    if (!c || !out_cls || !out_strand) return IOC_ERR_ARG;
    if (!c->have_params) return ioc_fail(c, IOC_ERR_STATE, "ioc_set_params first");
    const int n = c->n;
    std::vector<uint8_t> gated(size_t(n) + 1, 0);
    // thresholds live on the device; the tie replay needs them on the host: fetched once per set of queries
    if (c->h_min_total_gen != c->query_gen || c->h_min_total.size() != size_t(n) + 1) {
        c->h_min_total.assign(size_t(n) + 1, 0);
        if (n > 0) {
            hipError_t e = hipMemcpy(c->h_min_total.data(), c->d_min_total, size_t(n) * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, hipGetErrorString(e));
        }
        c->h_min_total_gen = c->query_gen;
    }
    const std::vector<uint32_t>& need = c->h_min_total;
    // every call is a pass over a batch of its own as far as the timings go: the table of the gap bounds (a function of the
    // queries, kept per query set by ioc_score) is computed again, as it would be for a batch seen for the first time
    c->gap_bound_gen = ~0ull;
    if (c->params.mode == IOC_MODE_FURIOUS || c->params.mode == IOC_MODE_NONE)
        return ioc_fail(c, IOC_ERR_STATE, "ioc_cluster_resident runs fast and sahlin mode: use ioc_cluster_batch for furious / none");
    if (c->params.mode == IOC_MODE_SAHLIN) {
        if (!c->have_res_seq || c->L > 0)
            return ioc_fail(c, IOC_ERR_STATE, "sahlin on resident queries needs ioc_resident_set_sequences (and no left clusters)");
        SeqAccess sa;
        sa.r_seq = c->res_seq.data();
        sa.r_off = c->res_off.data();
        sa.r_err = c->res_err.data();
        return run_pipeline(c, gated, need, out_cls, out_strand, stats, &sa);
    }
    return run_pipeline(c, gated, need, out_cls, out_strand, stats, nullptr);
}

}  // extern "C"
