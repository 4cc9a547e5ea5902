// ioc_consensus.cpp — ClusterSortedReads with the consensus branch on (src/cluster.cpp:200-204, 263-309;
// src/consensus.cpp:34-137): SURVEY.md §8 f4.
//
// With ConsMaxSize > 0 a cluster's representative is REPLACED by a consensus after (almost) every join once its
// graph holds ConsMinSize sequences, which breaks the decision-independence the parallel resolve rests on
//  — but only at those events.  The driver therefore speculates: the device pipeline
// (index build, scoring, resolve, alignment fallback) runs over ALL remaining entries against the current
// left state; the host walks the decisions in the reference's order, doing its bookkeeping, until the first
// join that replaces a representative; decisions up to there are final, everything after is recomputed
// against the updated state.  The representative's new minimizers come from the GPU extractor (K1), the
// index edit is UpdateMinDB.
//
// spoa is absent from the reference tree: the partial-order graphs stay on the caller's side, behind the five
// operations the reference performs on them (ioc_consensus_ops), the way parasail can stay behind
// ioc_get_ties / ioc_set_aln_verdicts.  Nothing here links the oracle; without a device every call fails.
#include <algorithm>
#include <iterator>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <memory>
#include <vector>

#include "ioc_internal.h"

namespace {

// a graph operation of the caller failed: keep what it left in the context's error text (the shipped POA engine reports
// there), the caller of the ABI sees both
int hook_fail(ioc_ctx* c, const char* what)
{
    const std::string inner = c->err;
    return ioc_fail(c, IOC_ERR_INPUT, std::string("consensus hook: ") + what + " failed" + (inner.empty() ? "" : ": " + inner));
}

struct ClState {
    uint64_t seq_id = 0;             // identity of the representative's sequence (alignment results are kept by identity)
    double raw_err = 0, hpc_err = 0;
    int64_t size = 0;                // cls[c]->size(): representative copy + members
    std::vector<uint32_t> vals;      // sorted distinct forward minimizer values of the representative
    std::string raw;                 // representative's raw sequence (alignment fallback / ConsPurge)
    bool have_raw = false;
};

void sorted_unique(std::vector<uint32_t>& v)
{
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
}

// UpdateMinDB's removal (minimizer.cpp:143-147): the list sorted, made unique, without the cluster.  The lists are strictly
// ascending already unless a caller loaded something else: then the one entry is cut out in place.
void drop_cluster(std::vector<uint32_t>& lst, uint32_t cls)
{
    bool ascending = true;
    for (size_t x = 1; x < lst.size() && ascending; ++x) ascending = lst[x - 1] < lst[x];
    if (ascending) {
        auto it = std::lower_bound(lst.begin(), lst.end(), cls);
        if (it != lst.end() && *it == cls) lst.erase(it);
        return;
    }
    sorted_unique(lst);
    lst.erase(std::remove(lst.begin(), lst.end(), cls), lst.end());
}

}  // namespace

// Values of the clusters whose representative changed during the current pass (old and new minimizer sets):
// a later entry of the pass keeps the decision the device made iff it shares fewer than
// int(MinShared * MinFraction) values with each of them — then none of them was, or becomes, a candidate that
// GetBestCluster looks at (cluster.cpp:324-406 walks candidates down to int(top * MinFraction), top >= MinShared).
struct DirtyIndex {
    static constexpr uint32_t CAP = 1u << 20, MASK = CAP - 1, EMPTY = 0xFFFFu;
    static constexpr int MAX_SLOTS = 256;  // clusters whose representative changes in one pass (deferred consensus batches that many events)
    std::vector<uint32_t> key = std::vector<uint32_t>(CAP, 0);
    std::vector<uint16_t> val = std::vector<uint16_t>(CAP, uint16_t(EMPTY));
    std::vector<uint32_t> touched;
    int nslots = 0;
    static uint32_t hash(uint32_t v) { return (v * 2654435761u) >> 12; }
    void reset()
    {
        for (uint32_t h : touched) val[h] = uint16_t(EMPTY);
        touched.clear();
        nslots = 0;
    }
    bool full() const { return nslots >= MAX_SLOTS || touched.size() > CAP / 4; }
    void add_cluster(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b)  // sorted, unique
    {
        std::vector<uint32_t> u;
        std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(u));
        for (uint32_t v : u) {
            uint32_t h = hash(v) & MASK;
            while (val[h] != EMPTY) h = (h + 1) & MASK;
            key[h] = v;
            val[h] = uint16_t(nslots);
            touched.push_back(h);
        }
        ++nslots;
    }
    // does the entry with these minimizer values (both strands' lists) reach `thr` shared values with any of them?
    bool touches(const uint32_t* v1, int64_t n1, const uint32_t* v2, int64_t n2, int thr) const
    {
        if (nslots == 0) return false;
        int cnt[MAX_SLOTS] = {0};
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t* v = pass ? v2 : v1;
            const int64_t nv = pass ? n2 : n1;
            for (int64_t x = 0; x < nv; ++x) {
                uint32_t h = hash(v[x]) & MASK;
                while (val[h] != EMPTY) {
                    if (key[h] == v[x] && ++cnt[val[h]] >= thr) return true;
                    h = (h + 1) & MASK;
                }
            }
        }
        return false;
    }
};

extern "C" {

int ioc_cluster_consensus(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_left_view* left,
                          const ioc_batch_view* rb, const ioc_consensus_args* ca, const ioc_consensus_ops* ops,
                          int32_t* out_cls, int8_t* out_strand, ioc_cluster_stats* stats)
{
    if (!c || !p || !table_path || !rb || !ca || !ops || !out_cls || !out_strand) return IOC_ERR_ARG;
    if (!ops->create || !ops->size || !ops->add || !ops->consensus || !ops->purge)
        return ioc_fail(c, IOC_ERR_ARG, "consensus needs all five graph operations");
    const int n = rb->n;
    if (n < 0) return ioc_fail(c, IOC_ERR_ARG, "negative batch size");
    c->err.clear();
    if (!rb->raw_seq || !rb->raw_off)
        return ioc_fail(c, IOC_ERR_ARG, "consensus needs the raw sequences of the right batch (graph seeds and additions)");
    const bool aln_mode = p->mode == IOC_MODE_SAHLIN || p->mode == IOC_MODE_FURIOUS;
    const int32_t L0 = left ? left->n_clusters : 0;
    if (left && left->n_keys < 0) return ioc_fail(c, IOC_ERR_ARG, "consensus takes the left MinDB as host arrays");
    if (aln_mode && L0 > 0 && (!left->rep_seq || !left->rep_off || !left->cls_raw_err))
        return ioc_fail(c, IOC_ERR_ARG, "sahlin/furious need the left representatives' sequences");

    // the windowed passes below read per-query state of every query of a pass: never sharded (ioc_set_shard)
    struct ShardOff {
        ioc_ctx* c;
        int world;
        explicit ShardOff(ioc_ctx* x) : c(x), world(x->shard_world) { c->shard_world = 1; }
        ~ShardOff() { c->shard_world = world; }
    } shard_off(c);
    struct DepSets {  // (run_pipeline leaves the order-dependent queries and their candidate sets while this driver runs)
        ioc_ctx* c;
        explicit DepSets(ioc_ctx* x) : c(x) { c->want_dep_sets = true; }
        ~DepSets() { c->want_dep_sets = false; }
    } dep_sets(c);
    // ---- left state on the host: MinDB as an ordered map, one ClState per cluster ----
    // (hashed: UpdateMinDB looks up hundreds of keys per event, among half a million; where the reference's std::map order
    // matters — the flat view, the export — the keys are sorted)
    std::unordered_map<uint32_t, std::vector<uint32_t>> db;
    if (left && left->n_keys > 0) db.reserve(size_t(left->n_keys) * 2);
    auto sorted_keys = [&](bool with_empty) {
        std::vector<uint32_t> ks;
        ks.reserve(db.size());
        for (auto& kv : db)
            if (with_empty || !kv.second.empty()) ks.push_back(kv.first);
        std::sort(ks.begin(), ks.end());
        return ks;
    };
    std::vector<ClState> cl(static_cast<size_t>(L0));
    // sequence identities: right entry i -> i; left representatives and consensus sequences -> n, n + 1, ...
    uint64_t next_seq_id = uint64_t(n);
    c->aln_cache.clear();
    struct IdsGuard {
        ioc_ctx* c;
        ~IdsGuard()
        {
            c->aln_qid.clear();
            c->aln_lid.clear();
            c->aln_cache.clear();
        }
    } ids_guard{c};
    for (int32_t t = 0; t < L0; ++t) {
        cl[size_t(t)].seq_id = next_seq_id++;
        cl[size_t(t)].hpc_err = left->cls_hpc_err[t];
        cl[size_t(t)].raw_err = left->cls_raw_err ? left->cls_raw_err[t] : 0.0;
        cl[size_t(t)].size = ca->left_sizes ? ca->left_sizes[t] : 2;
        if (left->rep_seq && left->rep_off) {
            cl[size_t(t)].raw.assign(left->rep_seq + left->rep_off[t], size_t(left->rep_off[t + 1] - left->rep_off[t]));
            cl[size_t(t)].have_raw = true;
        }
    }
    if (left)
        for (int64_t i = 0; i < left->n_keys; ++i) {
            auto& v = db[left->keys[i]];
            v.assign(left->postings + left->offs[i], left->postings + left->offs[i + 1]);
            for (uint32_t t : v) {
                if (t >= uint32_t(L0)) return ioc_fail(c, IOC_ERR_ARG, "left posting >= n_clusters");
                cl[t].vals.push_back(left->keys[i]);  // keys ascending -> vals come out sorted
            }
        }

    // layout of the right batch's minimizer lists: the usual "all forward lists, then all reverse lists" lets a
    // suffix of the batch be handed over by pointer arithmetic
    bool blocked = n > 0;
    for (int i = 0; i < n && blocked; ++i)
        blocked = rb->off_fwd[i] <= rb->off_fwd[i + 1] && rb->off_rev[i] <= rb->off_rev[i + 1];
    blocked = blocked && n > 0 && rb->off_fwd[n] <= rb->off_rev[0];
    if (n > 0 && !blocked)
        return ioc_fail(c, IOC_ERR_ARG, "consensus driver: minimizer lists must be laid out forward block, then reverse block");

    ioc_cluster_stats total{};
    std::vector<int64_t> of, orv, roff;
    std::vector<uint32_t> keys, post, keys2, post2, dirty_keys;
    std::vector<int64_t> offs, offs2;
    bool have_view = false;
    const bool view_check = getenv("IOC_CONS_VIEW_CHECK") != nullptr;
    std::vector<double> herr, rerr;
    std::string lseq;
    std::vector<int64_t> loff;
    std::vector<int32_t> sub_cls, sub_cut;
    // ALN_INVOKED (cluster.cpp:21, 559): an entry counts iff it reached the alignment fallback in the pass whose decision
    // stands for it — the flag of an entry is overwritten whenever a later pass walks it again
    std::vector<int32_t> sub_tgt;
    std::vector<int8_t> sub_str;
    std::vector<uint8_t> sub_flg, sub_dep, aln_flag(size_t(rb->n) + 1, 0);
    std::vector<std::vector<std::pair<int32_t, int8_t>>> sub_depset;  // per window entry with sub_dep: the candidates its hit order chooses among
    std::vector<int32_t> ncl_at;  // clusters that existed when the walk reached the window entry
    std::vector<int8_t> sub_strand;
    const int k = p->k, w = p->w;
    int pos = 0;
    // phase clock for IOC_TRACE: [0] left view, [1] device pass, [2] graph hooks, [3] new representative
    double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ([4] flush, [5] collect, [6] verification + UpdateMinDB, [7] rollback / commit)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // A pass decides a WINDOW of entries, not all that remain: a decision depends on earlier entries only, so a
    // prefix of the batch gives the same decisions, and everything behind the next consensus event would be thrown
    // away anyway.  The window follows the distance between events (fast mode: thousands of entries, sahlin mode
    // with small clusters: a handful).
    DirtyIndex dirty;
    const int dirty_thr = std::max(1, int(double(p->min_shared) * p->min_fraction));
    const bool one_event_per_pass = getenv("IOC_CONS_RESTART_ALWAYS") != nullptr;  // (the first version of this driver)
    int window = n;
    if (const char* e = getenv("IOC_CONS_WINDOW")) window = std::max(1, atoi(e));
    const bool fixed_window = getenv("IOC_CONS_WINDOW") != nullptr;
    std::vector<uint32_t> wval, wpos;
    // Deferred consensus (ioc_consensus_spec_ops; IOC_CONS_SPECULATE=0 switches it off): the walk does not wait for a
    // consensus — it records the event, queues the request and goes on as long as the entries it meets cannot see the
    // OLD representative of a cluster with a pending event.  At the end of the pass all requested consensus sequences
    // come back from ONE flush of the graph store (their additions aligned together), the new representatives are
    // re-minimized by ONE extractor launch, and the entries walked after each event are checked again, in order,
    // against old AND new minimizer sets — the reference's result is a function of those sets only.  An entry that
    // could see a new representative after all ends the pass there: host state is undone from the journal, the graph
    // store rolls back the operations tagged with later entries.
    const ioc_consensus_spec_ops* spec = ops->spec;
    if (const char* e = getenv("IOC_CONS_SPECULATE"))
        if (atoi(e) == 0) spec = nullptr;
    if (one_event_per_pass) spec = nullptr;
    struct PendingEvent {
        int x, i;
        int32_t dc;
        double hpc_err, raw_err;
    };
    struct Undo {
        int kind;  // 0 gated, 1 new cluster, 2 join
        int x;
        int32_t dc;
        int64_t dsize;
        std::vector<uint8_t> key_was_new;  // new cluster: per value, whether AddMinimizers opened the key
    };
    std::vector<PendingEvent> evs;
    std::vector<Undo> journal;
    std::vector<uint32_t> upd_keys;  // keys UpdateMinDB went through in the pass's verification (db[v] opens a key, even an empty one)
    DirtyIndex dirty_b;
    int64_t n_spec_rollbacks = 0, n_spec_events = 0, n_spec_flushes = 0;
    // IOC_CONS_FORCE_ROLLBACK=N (tests): every N-th entry looked at again is treated as if it could see a new
    // representative — a rollback only repeats work, the result must not change
    const int force_rb = getenv("IOC_CONS_FORCE_ROLLBACK") ? std::max(0, atoi(getenv("IOC_CONS_FORCE_ROLLBACK"))) : 0;
    int64_t rb_counter = 0;
    // The reference's hit order (GetMinimizerHits + SortMinimizerHits, minimizer.cpp:44-121) of entry i under the MinDB as it
    // stands, clusters below `ncl` only: the same unordered_map (hash, initial bucket count, insertion sequence: forward
    // minimizers in order, each posting list in order, then the reverse ones), the same std::sort.  The first candidate of `dep`
    // in that order is what an order-dependent decision comes to (§3 of DESIGN.md): the candidates themselves — clusters
    // whose representatives did not change — keep their Sizes and verdicts, only their order can move.
    auto host_order_pick = [&](int i, int32_t ncl, const std::vector<std::pair<int32_t, int8_t>>& dep, int32_t& w_cls, int8_t& w_strand) -> bool {
        typedef std::pair<int, int> SCl;
        struct SClHash {
            std::size_t operator()(const SCl& u) const { return size_t(int(u.first * u.second)); }
        };
        struct SHit {
            unsigned Size, Cls;
            int Strand;
        };
        const int64_t nf = rb->off_fwd[i + 1] - rb->off_fwd[i], nr = rb->off_rev[i + 1] - rb->off_rev[i];
        std::unordered_map<SCl, unsigned, SClHash> res(size_t(20) * size_t(nf + nr), SClHash());
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t* v = rb->min_val + (pass ? rb->off_rev[i] : rb->off_fwd[i]);
            const int64_t nv = pass ? nr : nf;
            for (int64_t x = 0; x < nv; ++x) {
                auto it = db.find(v[x]);
                if (it == db.end()) continue;
                for (uint32_t cid2 : it->second)
                    if (int32_t(cid2) < ncl) res[std::make_pair(int(cid2), pass ? -1 : 1)]++;
            }
        }
        std::vector<std::unique_ptr<SHit>> order;
        order.reserve(res.size());
        for (auto& kv : res) order.push_back(std::unique_ptr<SHit>(new SHit{kv.second, unsigned(kv.first.first), kv.first.second}));
        std::sort(order.begin(), order.end(), [](const std::unique_ptr<SHit>& a, const std::unique_ptr<SHit>& b) { return a->Size > b->Size; });
        for (auto& o : order)
            for (auto& d : dep)
                if (int32_t(o->Cls) == d.first && o->Strand == int(d.second)) {
                    w_cls = d.first;
                    w_strand = d.second;
                    return true;
                }
        return false;
    };
    while (pos < n) {
        const int m = std::min(n - pos, window);
        double t0 = now();
        // ---- left view of the current state ----
        const int32_t Lc = int32_t(cl.size());
        // The MinDB as flat arrays.  Walking the map (half a million keys, a heap vector behind each) cost 10 ms per pass; a
        // pass touches a few thousand keys, so the arrays of the previous pass are patched instead: untouched stretches are
        // copied, the keys written since (dirty_keys: AddMinimizers, UpdateMinDB; a rollback only touches keys of its own pass)
        // are looked up in the map.  IOC_CONS_VIEW_CHECK=1 builds it both ways and compares.
        auto full_view = [&](std::vector<uint32_t>& K, std::vector<int64_t>& O, std::vector<uint32_t>& P) {
            K.clear();
            O.clear();
            P.clear();
            K = sorted_keys(false);  // lists emptied by UpdateMinDB stay in the MinDB but match nothing
            for (uint32_t k2 : K) {
                const auto& lst = db.find(k2)->second;
                O.push_back(int64_t(P.size()));
                P.insert(P.end(), lst.begin(), lst.end());
            }
            O.push_back(int64_t(P.size()));
        };
        if (!have_view) {
            full_view(keys, offs, post);
            have_view = true;
        } else if (!dirty_keys.empty()) {
            std::sort(dirty_keys.begin(), dirty_keys.end());
            dirty_keys.erase(std::unique(dirty_keys.begin(), dirty_keys.end()), dirty_keys.end());
            keys2.clear();
            offs2.clear();
            post2.clear();
            keys2.reserve(keys.size() + dirty_keys.size());
            offs2.reserve(keys.size() + dirty_keys.size() + 1);
            post2.reserve(post.size() + post.size() / 16 + 4096);
            const size_t nk = keys.size();
            size_t a = 0;
            auto copy_range = [&](size_t from, size_t to) {  // entries [from, to) of the previous arrays, as they are
                if (from >= to) return;
                const int64_t delta = int64_t(post2.size()) - offs[from];
                keys2.insert(keys2.end(), keys.begin() + int64_t(from), keys.begin() + int64_t(to));
                const size_t o = offs2.size();
                offs2.resize(o + (to - from));
                for (size_t x = from; x < to; ++x) offs2[o + (x - from)] = offs[x] + delta;
                post2.insert(post2.end(), post.begin() + offs[from], post.begin() + offs[to]);
            };
            for (uint32_t dk : dirty_keys) {
                const size_t b = size_t(std::lower_bound(keys.begin() + int64_t(a), keys.end(), dk) - keys.begin());
                copy_range(a, b);
                a = b;
                if (a < nk && keys[a] == dk) ++a;
                auto it = db.find(dk);
                if (it != db.end() && !it->second.empty()) {
                    keys2.push_back(dk);
                    offs2.push_back(int64_t(post2.size()));
                    post2.insert(post2.end(), it->second.begin(), it->second.end());
                }
            }
            copy_range(a, nk);
            offs2.push_back(int64_t(post2.size()));
            keys.swap(keys2);
            offs.swap(offs2);
            post.swap(post2);
        }
        dirty_keys.clear();
        if (view_check) {
            full_view(keys2, offs2, post2);
            if (keys2 != keys || offs2 != offs || post2 != post) return ioc_fail(c, IOC_ERR_STATE, "consensus driver: the patched left view differs from a rebuilt one");
        }
        herr.resize(size_t(Lc));
        rerr.resize(size_t(Lc));
        c->aln_lid.resize(size_t(Lc));
        for (int32_t t = 0; t < Lc; ++t) {
            herr[size_t(t)] = cl[size_t(t)].hpc_err;
            rerr[size_t(t)] = cl[size_t(t)].raw_err;
            c->aln_lid[size_t(t)] = cl[size_t(t)].seq_id;
        }
        c->aln_qid.resize(size_t(m));
        for (int i = 0; i < m; ++i) c->aln_qid[size_t(i)] = uint64_t(pos + i);
        ioc_left_view lv{};
        lv.n_clusters = Lc;
        lv.cls_hpc_err = herr.data();
        lv.n_keys = int64_t(keys.size());
        lv.keys = keys.data();
        lv.offs = offs.data();
        lv.postings = post.data();
        if (aln_mode) {
            lseq.clear();
            loff.assign(size_t(Lc) + 1, 0);
            for (int32_t t = 0; t < Lc; ++t) {
                if (!cl[size_t(t)].have_raw) return ioc_fail(c, IOC_ERR_STATE, "a representative's sequence is missing");
                lseq += cl[size_t(t)].raw;
                loff[size_t(t) + 1] = int64_t(lseq.size());
            }
            lv.rep_seq = lseq.data();
            lv.rep_off = loff.data();
            lv.cls_raw_err = rerr.data();
        }
        // ---- the entries [pos, pos + m) as a batch view: their forward lists, then their reverse lists ----
        const int64_t fb = rb->off_fwd[pos], fe = rb->off_fwd[pos + m], vb = rb->off_rev[pos], ve = rb->off_rev[pos + m];
        const int64_t nf = fe - fb, nr = ve - vb;
        wval.resize(size_t(nf + nr) + 1);
        wpos.resize(size_t(nf + nr) + 1);
        if (nf) {
            memcpy(wval.data(), rb->min_val + fb, size_t(nf) * 4);
            memcpy(wpos.data(), rb->min_pos + fb, size_t(nf) * 4);
        }
        if (nr) {
            memcpy(wval.data() + nf, rb->min_val + vb, size_t(nr) * 4);
            memcpy(wpos.data() + nf, rb->min_pos + vb, size_t(nr) * 4);
        }
        of.resize(size_t(m) + 1);
        orv.resize(size_t(m) + 1);
        roff.resize(size_t(m) + 1);
        for (int i = 0; i <= m; ++i) {
            of[size_t(i)] = rb->off_fwd[pos + i] - fb;
            orv[size_t(i)] = nf + (rb->off_rev[pos + i] - vb);
            roff[size_t(i)] = rb->raw_off[pos + i] - rb->raw_off[pos];
        }
        ioc_batch_view sv = *rb;
        sv.n = m;
        sv.off_fwd = of.data();
        sv.off_rev = orv.data();
        sv.min_val = wval.data();
        sv.min_pos = wpos.data();
        sv.total = nf + nr;
        sv.raw_len = rb->raw_len + pos;
        sv.hpc_len = rb->hpc_len + pos;
        sv.score = rb->score + pos;
        sv.raw_err = rb->raw_err + pos;
        sv.hpc_err = rb->hpc_err + pos;
        sv.state = rb->state ? rb->state + pos : nullptr;
        sv.raw_seq = rb->raw_seq + rb->raw_off[pos];
        sv.raw_off = roff.data();
        sv.n_members = rb->n_members ? rb->n_members + pos : nullptr;
        sub_cls.assign(size_t(m) + 1, -1);
        sub_strand.assign(size_t(m) + 1, 0);
        ioc_cluster_stats st{};
        ph[0] += now() - t0;
        t0 = now();
        int r = ioc_cluster_merge(c, p, table_path, Lc > 0 ? &lv : nullptr, &sv, sub_cls.data(), sub_strand.data(), &st);
        if (r != IOC_OK) return r;
        sub_cut.assign(size_t(m) + 1, INT32_MAX);
        if (m > 0 && (r = ioc_get_cuts(c, sub_cut.data())) != IOC_OK) return r;
        sub_tgt.assign(size_t(m) + 1, 0);
        sub_str.assign(size_t(m) + 1, 0);
        sub_flg.assign(size_t(m) + 1, 0);
        if (m > 0 && (r = ioc_get_decisions(c, sub_tgt.data(), sub_str.data(), sub_flg.data())) != IOC_OK) return r;
        sub_dep.assign(size_t(m) + 1, 0);
        sub_depset.assign(size_t(m) + 1, std::vector<std::pair<int32_t, int8_t>>());
        for (int x = 0; x < m && size_t(x) < c->last_order_dep.size(); ++x) {
            sub_dep[size_t(x)] = c->last_order_dep[size_t(x)];
            if (sub_dep[size_t(x)] && size_t(x) < c->last_dep_set.size()) sub_depset[size_t(x)] = c->last_dep_set[size_t(x)];
        }
        ncl_at.assign(size_t(m) + 1, 0);
        ph[1] += now() - t0;
        total.resolve_iters += st.resolve_iters;
        total.n_tie_replays += st.n_tie_replays;
        total.aln_rounds += st.aln_rounds;
        total.n_aln_pairs += st.n_aln_pairs;
        total.n_cons_restarts++;
        if (getenv("IOC_TRACE")) fprintf(stderr, "[ioc] consensus pass from entry %d (%d left clusters)\n", pos, Lc);

        // ---- walk the decisions in the reference's order until a representative changes ----
        const int pass_from = pos;
        bool restarted = false;
        // Deferred mode walks a device pass in SEGMENTS: a segment ends — flush, new representatives, verification — where a pass
        // used to end, and also in front of an order-dependent entry met with consensus requests pending; in that case the walk
        // goes on from that entry on the same device decisions (dirty_b keeps every representative changed since the device pass:
        // an entry that can see one of them ends the device pass).
        dirty_b.reset();
        int x_begin = 0;
        bool next_pass = false;
        for (bool again = true; again;) {
        again = false;
        bool resume = false;
        dirty.reset();
        evs.clear();
        journal.clear();
        int stop_x = m;  // (deferred mode) the walk stands up to here unless the verification says otherwise
        for (int x = x_begin; x < m; ++x) {
            const int i = pos + x;
            int32_t dc = sub_cls[size_t(x)];
            ncl_at[size_t(x)] = int32_t(cl.size());
            aln_flag[size_t(i)] = dc >= 0 && (sub_flg[size_t(x)] & 2) ? 1 : 0;
            if (dc < 0) {  // (gated by its quality: no cluster has a say)
                out_cls[i] = -1;
                out_strand[i] = 0;
                total.n_gated++;
                if (spec && !evs.empty()) journal.push_back(Undo{0, x, -1, 0, {}});
                continue;
            }
            // Representatives changed earlier in this pass: the device's decision for this entry stands only if
            // the entry cannot see any of them — it shares fewer values with each than the Size its mapping walk
            // stops at (int(top * MinFraction), ioc_get_cuts; without a walk: what would start one).
            if (spec && dirty_b.nslots) {  // (representatives replaced in an earlier segment of this device pass)
                const int thr = sub_cut[size_t(x)] == INT32_MAX ? dirty_thr : std::max(dirty_thr, int(sub_cut[size_t(x)]));
                if (dirty_b.full() ||
                    dirty_b.touches(rb->min_val + rb->off_fwd[i], rb->off_fwd[i + 1] - rb->off_fwd[i], rb->min_val + rb->off_rev[i],
                                    rb->off_rev[i + 1] - rb->off_rev[i], (sub_dep[size_t(x)] && sub_depset[size_t(x)].empty()) ? 1 : thr)) {
                    stop_x = x;  // the device pass ends here
                    break;
                }
            }
            if (dirty.nslots) {
                const int thr = sub_cut[size_t(x)] == INT32_MAX ? dirty_thr : std::max(dirty_thr, int(sub_cut[size_t(x)]));
                // (a decision that hangs on the reference's hit ORDER — a tie at the top Size, several candidates that align —
                // depends on which (cluster, strand) keys the hit map holds at all and on their Sizes, down to Size 1: the
                // iteration order of the unordered_map and the path of its std::sort change with them.  ONE value shared with a
                // changed representative's old or new set can add, remove or resize such a key.  The candidates the order chooses
                // among are known (sub_depset): the order itself is computed again on the host below — at once when every
                // consensus so far has been taken, in the verification of the pass when they are deferred.)
                if (dirty.touches(rb->min_val + rb->off_fwd[i], rb->off_fwd[i + 1] - rb->off_fwd[i], rb->min_val + rb->off_rev[i],
                                  rb->off_rev[i + 1] - rb->off_rev[i], (sub_dep[size_t(x)] && sub_depset[size_t(x)].empty()) ? 1 : thr)) {
                    if (spec) {  // (it sees the OLD representative of a cluster with a pending event: the pass ends here)
                        stop_x = x;
                        break;
                    }
                    pos = i;
                    restarted = true;
                    break;
                }
                if (!spec && sub_dep[size_t(x)] && !sub_depset[size_t(x)].empty() && dc < int32_t(cl.size())) {
                    int32_t wc = -1;
                    int8_t ws = 0;
                    if (!host_order_pick(i, int32_t(cl.size()), sub_depset[size_t(x)], wc, ws))
                        return ioc_fail(c, IOC_ERR_STATE, "consensus driver: none of an order-dependent decision's candidates is a hit any more");
                    sub_cls[size_t(x)] = dc = wc;
                    sub_strand[size_t(x)] = ws;
                }
            }
            if (spec && sub_dep[size_t(x)] && !sub_depset[size_t(x)].empty() && dc < int32_t(cl.size())) {
                if (!evs.empty()) {  // requests pending: the segment ends in front of this entry, the walk goes on from it afterwards
                    stop_x = x;
                    resume = true;
                    break;
                }
                if (dirty_b.nslots) {  // nothing pending, the MinDB is exact: the order as it is now
                    int32_t wc = -1;
                    int8_t ws = 0;
                    if (!host_order_pick(i, int32_t(cl.size()), sub_depset[size_t(x)], wc, ws))
                        return ioc_fail(c, IOC_ERR_STATE, "consensus driver: none of an order-dependent decision's candidates is a hit any more");
                    sub_cls[size_t(x)] = dc = wc;
                    sub_strand[size_t(x)] = ws;
                }
            }
            const char* rseq = rb->raw_seq + rb->raw_off[i];
            const int rlen = int(rb->raw_off[i + 1] - rb->raw_off[i]);
            const int64_t entry_size = rb->n_members ? int64_t(rb->n_members[i]) + 1 : 1;  // reads[i]->size()
            if (dc == int32_t(cl.size())) {
                // ---- opens a new cluster (cluster.cpp:177-222) ----
                ClState ns;
                ns.seq_id = uint64_t(i);  // the representative is this read
                ns.raw_err = rb->raw_err[i];
                ns.hpc_err = rb->hpc_err[i];
                ns.size = entry_size == 1 ? 2 : entry_size;  // a fresh read gets a representative copy in front
                ns.vals.assign(rb->min_val + rb->off_fwd[i], rb->min_val + rb->off_fwd[i + 1]);
                sorted_unique(ns.vals);
                ns.raw.assign(rseq, size_t(rlen));
                ns.have_raw = true;
                // AddMinimizers (minimizer.cpp:31-42): the new id is larger than every id in the lists
                if (spec && !evs.empty()) {
                    Undo u{1, x, dc, 0, {}};
                    u.key_was_new.reserve(ns.vals.size());
                    for (uint32_t v : ns.vals) u.key_was_new.push_back(db.find(v) == db.end() ? 1 : 0);
                    journal.push_back(std::move(u));
                }
                for (uint32_t v : ns.vals) db[v].push_back(uint32_t(dc));
                dirty_keys.insert(dirty_keys.end(), ns.vals.begin(), ns.vals.end());
                // (before the first event of a pass nothing is ever undone: no snapshot needed)
                if ((spec && !evs.empty() ? spec->create_tagged(ops->user, 0, dc, rseq, rlen, i) : ops->create(ops->user, 0, dc, rseq, rlen)) < 0)
                    return hook_fail(c, "create");
                cl.push_back(std::move(ns));
                out_cls[i] = dc;
                out_strand[i] = 1;
                continue;
            }
            if (dc > int32_t(cl.size())) return ioc_fail(c, IOC_ERR_STATE, "inconsistent cluster id from the device path");
            // ---- joins cluster dc (cluster.cpp:223-309) ----
            ClState& b = cl[size_t(dc)];
            out_cls[i] = dc;
            out_strand[i] = sub_strand[size_t(x)];
            total.n_joined++;
            b.size += entry_size > 1 ? entry_size - 1 : 1;
            if (spec && !evs.empty()) journal.push_back(Undo{2, x, dc, entry_size > 1 ? entry_size - 1 : 1, {}});
            if (ca->cons_max_size <= 0) continue;
            if (ca->left_depth == -1 && ca->cons_period > 0 && b.size > ca->cons_period) continue;   // :267-271
            const int cons_min = ca->left_depth != -1 ? 2 : ca->cons_min_size;                         // :284-288
            // UpdateClusterConsensus, consensus.cpp:34-126
            const int left_size = ops->size(ops->user, 0, dc);
            if (left_size < 0) return ioc_fail(c, IOC_ERR_INPUT, "consensus hook: the cluster has no graph");
            const int rsz = ops->size(ops->user, 1, i);
            const bool have_right = rsz >= 0;
            const int right_size = have_right ? rsz : 1;
            const double hpc_err = (b.hpc_err * double(left_size) + rb->hpc_err[i] * double(right_size)) / double(left_size + right_size);
            const double raw_err = (b.raw_err * double(left_size) + rb->raw_err[i] * double(right_size)) / double(left_size + right_size);
            // (the reference reverse-complements a copy and throws it away, consensus.cpp:47-49: the read goes in as it is)
            t0 = now();
            if ((spec ? spec->add_tagged(ops->user, 0, dc, rseq, rlen, have_right ? unsigned(right_size) : 1u, i)
                      : ops->add(ops->user, 0, dc, rseq, rlen, have_right ? unsigned(right_size) : 1u)) < 0)
                return hook_fail(c, "add");
            ph[2] += now() - t0;
            if (ops->size(ops->user, 0, dc) < cons_min) continue;
            if (spec) {
                // ---- the consensus is requested, not awaited ----
                if (spec->consensus_deferred(ops->user, 0, dc, i) < 0) return hook_fail(c, "deferred consensus");
                if (evs.empty()) journal.clear();
                evs.push_back(PendingEvent{x, i, dc, hpc_err, raw_err});
                dirty.add_cluster(b.vals, std::vector<uint32_t>());  // what later entries must not see: the OLD set for now
                n_spec_events++;
                if (dirty.full()) {
                    stop_x = x + 1;
                    break;
                }
                continue;
            }
            t0 = now();
            std::vector<char> buf(size_t(1) << 22);
            const int clen = ops->consensus(ops->user, 0, dc, buf.data(), int(buf.size()));
            if (clen < 0) return hook_fail(c, "consensus");
            ph[2] += now() - t0;
            t0 = now();
            std::string cons(buf.data(), size_t(clen));
            // the new representative: fixed quality character, HPC, minimizers (K1 on the GPU)
            const char qraw = std::to_string(int(-10 * log10(raw_err)) + 33)[0];  // :98-99: first CHARACTER of the number
            if (!(cons.size() > size_t(2 * k) || cons.size() >= size_t(w)))
                return ioc_fail(c, IOC_ERR_INPUT, "consensus shorter than 2k and w (the reference re-minimizes an empty sequence here)");
            const std::string qual(cons.size(), qraw);
            const int64_t xoff[2] = {0, int64_t(cons.size())};
            uint32_t hlen = 0;
            double herr_k1 = 0;
            int64_t xf[2] = {0, 0}, xr[2] = {0, 0};
            int32_t xst = 0;
            r = ioc_extract_minimizers(c, 1, xoff, reinterpret_cast<const uint8_t*>(cons.data()),
                                       reinterpret_cast<const uint8_t*>(qual.data()), k, w, &hlen, &herr_k1, xf, xr, &xst);
            if (r != IOC_OK) return r;
            if (xst != 0)
                return ioc_fail(c, IOC_ERR_INPUT, "consensus with a non-ACGT base or an HPC length below 2k / w");
            const int64_t nmin = xr[1];
            std::vector<uint32_t> mv(size_t(nmin) + 1), mp(size_t(nmin) + 1);
            if ((r = ioc_extracted_download(c, mv.data(), mp.data(), nmin)) != IOC_OK) return r;
            std::vector<char> hs(cons.size() + 1), hq(cons.size() + 1);
            if ((r = ioc_extracted_hpc_download(c, hs.data(), hq.data(), int64_t(cons.size()))) != IOC_OK) return r;
            // UpdateMinDB (minimizer.cpp:124-160) on the host MinDB
            std::vector<uint32_t> nv(mv.begin() + xf[0], mv.begin() + xf[1]);
            sorted_unique(nv);
            {
                std::vector<uint32_t> to_del, to_ins;
                std::set_difference(b.vals.begin(), b.vals.end(), nv.begin(), nv.end(), std::back_inserter(to_del));
                std::set_difference(nv.begin(), nv.end(), b.vals.begin(), b.vals.end(), std::back_inserter(to_ins));
                for (uint32_t v : to_del) {
                    auto& lst = db[v];
                    drop_cluster(lst, uint32_t(dc));
                }
                for (uint32_t v : to_ins) {
                    auto& lst = db[v];
                    lst.push_back(uint32_t(dc));
                    std::sort(lst.begin(), lst.end());
                }
                dirty_keys.insert(dirty_keys.end(), to_del.begin(), to_del.end());
                dirty_keys.insert(dirty_keys.end(), to_ins.begin(), to_ins.end());
            }
            dirty.add_cluster(b.vals, nv);
            b.vals.swap(nv);
            b.raw_err = raw_err;
            b.hpc_err = hpc_err;  // consensus.cpp:121 — also when the 0.9999 branch (:112-117) fired
            b.raw = cons;
            b.have_raw = true;
            b.seq_id = next_seq_id++;
            total.n_cons_invoked++;
            if (ops->rep_changed) {
                ioc_rep_record rec{};
                rec.raw_seq = cons.data();
                rec.raw_len = int32_t(cons.size());
                rec.raw_qual = qraw;
                rec.raw_err = raw_err;
                rec.raw_score = raw_err * double(cons.size());
                rec.hpc_seq = hs.data();
                rec.hpc_len = int32_t(hlen);
                rec.hpc_err = hpc_err;
                rec.fwd_min = mv.data() + xf[0];
                rec.fwd_pos = mp.data() + xf[0];
                rec.n_fwd = int32_t(xf[1] - xf[0]);
                rec.rev_min = mv.data() + xr[0];
                rec.rev_pos = mp.data() + xr[0];
                rec.n_rev = int32_t(xr[1] - xr[0]);
                rec.entry = i;
                ops->rep_changed(ops->user, dc, &rec);
            }
            const int gsz = ops->size(ops->user, 0, dc);
            if (gsz > ca->cons_max_size) {  // ConsPurge, consensus.cpp:128-137
                if (ops->purge(ops->user, 0, dc, cons.data(), int(cons.size()), unsigned(gsz)) < 0)
                    return hook_fail(c, "purge");
            }
            ph[3] += now() - t0;
            // every later entry that can see this cluster has to see the new representative: the walk goes on
            // until it meets one (the check at the top), or restarts here when too many clusters have changed
            if (one_event_per_pass || dirty.full()) {
                pos = i + 1;
                restarted = true;
                break;
            }
        }
        if (spec) {
            // ---- deferred mode, second half of the pass: consensus sequences, new representatives, verification ----
            const int pos0 = pass_from;
            int end_x = stop_x;  // entries [0, end_x) of the window stand
            if (!evs.empty()) {
                double t1 = now();
                n_spec_flushes++;
                if (spec->flush(ops->user, evs[0].i) < 0) return hook_fail(c, "flush");
                ph[4] += now() - t1;
                const size_t ne = evs.size();
                std::vector<std::string> cons(ne);
                std::vector<char> buf(size_t(1) << 22);
                for (size_t e = 0; e < ne; ++e) {
                    const int clen = spec->collect(ops->user, 0, evs[e].dc, evs[e].i, buf.data(), int(buf.size()));
                    if (clen < 0) return hook_fail(c, "consensus");
                    cons[e].assign(buf.data(), size_t(clen));
                    if (!(cons[e].size() > size_t(2 * k) || cons[e].size() >= size_t(w)))
                        return ioc_fail(c, IOC_ERR_INPUT, "consensus shorter than 2k and w (the reference re-minimizes an empty sequence here)");
                }
                ph[2] += now() - t1;
                ph[5] += now() - t1;
                t1 = now();
                // the new representatives: fixed quality character, HPC, minimizers — ONE extractor call for all of them
                std::vector<int64_t> xoff(ne + 1, 0);
                std::string xseq, xqual;
                std::vector<char> qraw(ne);
                for (size_t e = 0; e < ne; ++e) {
                    qraw[e] = std::to_string(int(-10 * log10(evs[e].raw_err)) + 33)[0];  // :98-99: first CHARACTER of the number
                    xseq += cons[e];
                    xqual.append(cons[e].size(), qraw[e]);
                    xoff[e + 1] = int64_t(xseq.size());
                }
                std::vector<uint32_t> hlen(ne);
                std::vector<double> herr_k1(ne);
                std::vector<int64_t> xf(ne + 1), xr(ne + 1);
                std::vector<int32_t> xst(ne);
                int r2 = ioc_extract_minimizers(c, int32_t(ne), xoff.data(), reinterpret_cast<const uint8_t*>(xseq.data()),
                                                reinterpret_cast<const uint8_t*>(xqual.data()), k, w, hlen.data(), herr_k1.data(), xf.data(),
                                                xr.data(), xst.data());
                if (r2 != IOC_OK) return r2;
                for (size_t e = 0; e < ne; ++e)
                    if (xst[e] != 0) return ioc_fail(c, IOC_ERR_INPUT, "consensus with a non-ACGT base or an HPC length below 2k / w");
                const int64_t nmin = xr[ne];
                std::vector<uint32_t> mv(size_t(nmin) + 1), mp(size_t(nmin) + 1);
                if ((r2 = ioc_extracted_download(c, mv.data(), mp.data(), nmin)) != IOC_OK) return r2;
                std::vector<char> hs(xseq.size() + 1), hq(xseq.size() + 1);
                if ((r2 = ioc_extracted_hpc_download(c, hs.data(), hq.data(), int64_t(xseq.size()))) != IOC_OK) return r2;
                ph[3] += now() - t1;
                t1 = now();
                // ---- in the reference's order: finalize event e, then look again at the entries walked after it ----
                upd_keys.clear();
                size_t e = 0;
                int violation = -1;
                for (int x = evs[0].x; x < stop_x && violation < 0; ++x) {
                    const int i = pos0 + x;
                    if (force_rb && x > evs[0].x && (++rb_counter % force_rb) == 0) {
                        violation = x;
                        break;
                    }
                    if (x > evs[0].x && sub_cls[size_t(x)] >= 0 && dirty_b.nslots) {
                        // (an entry with an event of its own is looked at again like any other, before its event counts)
                        const int thr = sub_cut[size_t(x)] == INT32_MAX ? dirty_thr : std::max(dirty_thr, int(sub_cut[size_t(x)]));
                        if (dirty_b.touches(rb->min_val + rb->off_fwd[i], rb->off_fwd[i + 1] - rb->off_fwd[i], rb->min_val + rb->off_rev[i],
                                            rb->off_rev[i + 1] - rb->off_rev[i], (sub_dep[size_t(x)] && sub_depset[size_t(x)].empty()) ? 1 : thr)) {
                            violation = x;  // it can see a NEW representative: everything from here on is decided again
                            break;
                        }
                        if (sub_dep[size_t(x)] && !sub_depset[size_t(x)].empty() && sub_cls[size_t(x)] < ncl_at[size_t(x)]) {
                            // (an order-dependent decision: the order under the MinDB as the events before this entry leave it)
                            int32_t wc = -1;
                            int8_t ws = 0;
                            if (!host_order_pick(i, ncl_at[size_t(x)], sub_depset[size_t(x)], wc, ws) || wc != sub_cls[size_t(x)] ||
                                ws != sub_strand[size_t(x)]) {
                                violation = x;
                                break;
                            }
                        }
                    }
                    if (e < ne && evs[e].x == x) {
                        const PendingEvent& ev = evs[e];
                        ClState& b = cl[size_t(ev.dc)];
                        std::vector<uint32_t> nv(mv.begin() + xf[e], mv.begin() + xf[e + 1]);
                        sorted_unique(nv);
                        {   // UpdateMinDB (minimizer.cpp:124-160) on the host MinDB
                            std::vector<uint32_t> to_del, to_ins;
                            std::set_difference(b.vals.begin(), b.vals.end(), nv.begin(), nv.end(), std::back_inserter(to_del));
                            std::set_difference(nv.begin(), nv.end(), b.vals.begin(), b.vals.end(), std::back_inserter(to_ins));
                            for (uint32_t v : to_del) {
                                auto& lst = db[v];
                                drop_cluster(lst, uint32_t(ev.dc));
                                upd_keys.push_back(v);
                                dirty_keys.push_back(v);
                            }
                            for (uint32_t v : to_ins) {
                                auto& lst = db[v];
                                lst.push_back(uint32_t(ev.dc));
                                std::sort(lst.begin(), lst.end());
                                upd_keys.push_back(v);
                                dirty_keys.push_back(v);
                            }
                        }
                        dirty_b.add_cluster(b.vals, nv);
                        b.vals.swap(nv);
                        b.raw_err = ev.raw_err;
                        b.hpc_err = ev.hpc_err;  // consensus.cpp:121 — also when the 0.9999 branch (:112-117) fired
                        b.raw = cons[e];
                        b.have_raw = true;
                        b.seq_id = next_seq_id++;
                        total.n_cons_invoked++;
                        if (ops->rep_changed) {
                            ioc_rep_record rec{};
                            rec.raw_seq = cons[e].data();
                            rec.raw_len = int32_t(cons[e].size());
                            rec.raw_qual = qraw[e];
                            rec.raw_err = ev.raw_err;
                            rec.raw_score = ev.raw_err * double(cons[e].size());
                            rec.hpc_seq = hs.data() + xoff[e];
                            rec.hpc_len = int32_t(hlen[e]);
                            rec.hpc_err = ev.hpc_err;
                            rec.fwd_min = mv.data() + xf[e];
                            rec.fwd_pos = mp.data() + xf[e];
                            rec.n_fwd = int32_t(xf[e + 1] - xf[e]);
                            rec.rev_min = mv.data() + xr[e];
                            rec.rev_pos = mp.data() + xr[e];
                            rec.n_rev = int32_t(xr[e + 1] - xr[e]);
                            rec.entry = ev.i;
                            ops->rep_changed(ops->user, ev.dc, &rec);
                        }
                        const int gsz = ops->size(ops->user, 0, ev.dc);
                        if (gsz > ca->cons_max_size) {  // ConsPurge, consensus.cpp:128-137
                            if (ops->purge(ops->user, 0, ev.dc, cons[e].data(), int(cons[e].size()), unsigned(gsz)) < 0)
                                return hook_fail(c, "purge");
                        }
                        ++e;
                        if (dirty_b.full() && x + 1 < stop_x) violation = x + 1;  // too many changed clusters to keep checking: the pass ends here
                        continue;
                    }
                }
                ph[6] += now() - t1;
                const double t2 = now();
                if (violation >= 0) {
                    // ---- undo what the walk did for the entries [violation, stop_x), newest first ----
                    n_spec_rollbacks++;
                    std::sort(upd_keys.begin(), upd_keys.end());
                    for (size_t u = journal.size(); u-- > 0;) {
                        const Undo& un = journal[u];
                        if (un.x < violation) break;
                        if (un.kind == 0) {
                            total.n_gated--;
                        } else if (un.kind == 1) {
                            ClState& ns = cl.back();
                            if (int32_t(cl.size()) - 1 != un.dc) return ioc_fail(c, IOC_ERR_STATE, "consensus rollback: cluster stack out of order");
                            for (size_t y = ns.vals.size(); y-- > 0;) {
                                auto it = db.find(ns.vals[y]);
                                if (it == db.end() || it->second.empty() || it->second.back() != uint32_t(un.dc))
                                    return ioc_fail(c, IOC_ERR_STATE, "consensus rollback: MinDB out of order");
                                it->second.pop_back();
                                // the key goes with the cluster that opened it — unless an event that stands went through it
                                // since (in the reference's order UpdateMinDB's db[v] would have opened it, minimizer.cpp:143-152)
                                if (un.key_was_new[y] && it->second.empty() && !std::binary_search(upd_keys.begin(), upd_keys.end(), ns.vals[y]))
                                    db.erase(it);
                            }
                            cl.pop_back();
                        } else {
                            cl[size_t(un.dc)].size -= un.dsize;
                            total.n_joined--;
                        }
                    }
                    if (spec->rollback(ops->user, pos0 + violation) < 0) return hook_fail(c, "rollback");
                    end_x = violation;
                } else {
                    if (spec->commit(ops->user) < 0) return hook_fail(c, "commit");
                }
                ph[3] += now() - t1;
                ph[7] += now() - t2;
            } else if (spec->commit(ops->user) < 0) {
                return hook_fail(c, "commit");
            }
            if (getenv("IOC_TRACE") && !evs.empty())
                fprintf(stderr, "[ioc]   deferred: %zu events in the segment, entries [%d, %d) stand\n", evs.size(), pass_from + x_begin, pos0 + end_x);
            if (resume && end_x == stop_x && !dirty_b.full()) {  // (no violation: on from the order-dependent entry)
                x_begin = stop_x;
                again = true;
                continue;
            }
            pos = pos0 + end_x;
            restarted = end_x < m;
            if (restarted) {
                if (!fixed_window) window = std::max(64, 4 * std::max(1, pos - pass_from));
            } else if (!fixed_window) {
                window = std::min(n, std::max(64, 2 * window));
            }
            next_pass = true;
        }
        }  // segments
        if (next_pass) continue;
        if (restarted) {
            // the next event is probably as far away as this one was
            if (!fixed_window) window = std::max(64, 4 * std::max(1, pos - pass_from));
        } else {
            pos += m;  // the whole window stands
            if (!fixed_window) window = std::min(n, std::max(64, 2 * window));
        }
    }
    if (getenv("IOC_TRACE") && spec)
        fprintf(stderr, "[ioc] deferred consensus: %lld events in %lld flushes, %lld rollbacks\n", (long long)n_spec_events, (long long)n_spec_flushes,
                (long long)n_spec_rollbacks);
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc] consensus phases: left view %.1f ms, device passes %.1f ms, graph hooks %.1f ms (flush %.1f, flush + collect %.1f), new representatives %.1f ms "
                        "(verification + UpdateMinDB %.1f, rollback / commit %.1f)\n",
                ph[0], ph[1], ph[2], ph[4], ph[5], ph[3], ph[6], ph[7]);
    // the final MinDB is what ioc_index_export returns
    c->exp_keys.clear();
    c->exp_offs.clear();
    c->exp_post.clear();
    for (uint32_t k2 : sorted_keys(true)) {
        const auto& lst = db.find(k2)->second;
        c->exp_keys.push_back(k2);
        c->exp_offs.push_back(int64_t(c->exp_post.size()));
        c->exp_post.insert(c->exp_post.end(), lst.begin(), lst.end());
    }
    c->exp_offs.push_back(int64_t(c->exp_post.size()));
    c->exp_valid = true;
    c->resolved = true;
    // the device state belongs to this driver's LAST windowed pass: a later ioc_set_aln_verdicts + ioc_resolve on the
    // context must not warm-start from it
    c->warm_first = -1;
    total.n_clusters = int64_t(cl.size());
    total.n_aln_invoked = 0;
    for (int i = 0; i < rb->n; ++i) total.n_aln_invoked += aln_flag[size_t(i)];
    if (stats) *stats = total;
    return IOC_OK;
}

}  // extern "C"
