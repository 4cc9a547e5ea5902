/*
 * oracle.cpp — CPU oracle for the isONclust2 read->cluster assignment hot path.
 *
 *   *** TEST INFRASTRUCTURE.  Not product code, never a fallback. ***
 *
 * A from-scratch restatement of the reference algorithm; every function cites the
 * reference file:line it follows (paths relative to /root/reference).  The selection path
 * deliberately uses the same libstdc++ containers as the reference (unordered_map with the
 * same hash functors and bucket hints, std::sort over unique_ptr<SortedHit>) because the
 * order of equal-Size candidates is defined by them (src/cluster.cpp:622-636).
 *
 * Parity pin: tests/test_oracle_golden.py checks it against the reference's own unit-test
 * known answers (test/isONclust2_test.cpp) and against oracle/_ref (the reference TUs that
 * compile stand-alone: kmer_index.cpp, util.cpp, p_emp_prob.cpp).
 *
 * Third-party arithmetic absent from /root/reference: parasail (jeffdaily/parasail, .gitmodules:4-6, pinned
 * version unrecoverable) and spoa (src/consensus.cpp).  For the alignment fallback (src/cluster.cpp:408-515)
 * the oracle carries its OWN scalar statement of the call the reference makes — parasail_sg_trace_scan +
 * parasail_result_get_traceback, published semantics: semi-global, all four ends free, a gap of length n costs
 * open + (n-1)*extend, the comparison string spans the whole alignment incl. the end gaps (`sg_trace` below;
 * tie-breaking PARITY UNPINNED beyond the reference's AlnRatioTest vector).  It is independent of the product's
 * aligners: nothing under isonclust2_amd/ is called.  orc_set_aligner can still put another aligner behind
 * the same seam (tests use it to cross-check the product's host aligner against this one).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <iterator>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

// ---- src/minimizer.h:20-29 ------------------------------------------------------------
struct Mz {
    unsigned Min, Pos, Index;
};
typedef std::vector<Mz> MzVec;

// ---- src/minimizer.h:37-42, 52-58 ------------------------------------------------------
struct IdHash {
    std::size_t operator()(const unsigned& u) const { return std::size_t(u); }
};
typedef std::pair<int, int> SCl;
struct SClHash {
    std::size_t operator()(const SCl& u) const { return size_t(int(u.first * u.second)); }
};
typedef std::unordered_map<unsigned, std::vector<unsigned>, IdHash> MinDB;  // minimizer.h:60-61
struct Hit {
    unsigned Pos, Index;
};
struct RawHit {
    unsigned Cls;
    Hit H;
};
typedef std::vector<Hit> HitVec;
typedef std::unordered_map<SCl, HitVec, SClHash> HitMap;  // minimizer.h:75-76
struct SortedHit {                                        // minimizer.h:84-91
    unsigned Size, Cls;
    int Strand;
};
typedef std::vector<std::unique_ptr<SortedHit>> SortedHits;

// ---- src/kmer_index.h:29-45 --------------------------------------------------------------
inline unsigned base_code(char c)
{
    if (c == 'A') return 0;
    if (c == 'C') return 1;
    if (c == 'G') return 2;
    if (c == 'T') return 3;
    return unsigned(-1);
}

// ---- src/kmer_index.cpp:5-17 + kmer_index.h:59-66 -------------------------------------------
// value = sum base*4^(k-1-j) in 32-bit unsigned arithmetic; emits len-k k-mers (the final
// k-mer is never produced: loop bound `i < len - k`); empty if len < k.
std::vector<unsigned> kmer_encode(const std::string& s, unsigned k)
{
    std::vector<unsigned> out;
    if (s.length() < k) return out;
    size_t n = s.length() - k;
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        unsigned v = 0;
        for (unsigned j = 0; j < k; j++) v = 4u * v + base_code(s[i + j]);
        out.push_back(v);
    }
    return out;
}

// ---- src/minimizer.cpp:78-123 ------------------------------------------------------------
// Window of w-k+1 consecutive k-mers.  First window: leftmost minimum.  Each slide: if the
// value leaving equals the tracked minimum -> rescan (leftmost min) and emit; else if the
// entering value is strictly smaller -> emit it.  The reference reads kmerSeq[0..w-k]
// unguarded (UB when n <= w-k); the oracle returns an empty list there.
MzVec minimizers(const std::vector<unsigned>& ks, int k, int w)
{
    MzVec out;
    int initW = w - k;
    int n = int(ks.size());
    if (n <= initW || initW < 0) return out;
    out.reserve(size_t(n - initW));
    std::deque<unsigned> win;
    for (int i = 0; i <= initW; i++) win.push_back(ks[size_t(i)]);
    auto it = std::min_element(win.begin(), win.end());
    unsigned cur = *it;
    unsigned idx = 0;
    out.push_back(Mz{cur, unsigned(std::distance(win.begin(), it)), idx++});
    for (int i = initW + 1; i < n; i++) {
        unsigned nw = ks[size_t(i)];
        unsigned old = win.front();
        win.pop_front();
        win.push_back(nw);
        if (cur == old) {
            it = std::min_element(win.begin(), win.end());
            unsigned pos = unsigned(std::distance(win.begin(), it) + i - initW);
            cur = *it;
            out.push_back(Mz{cur, pos, idx++});
        } else if (nw < cur) {
            cur = nw;
            out.push_back(Mz{nw, unsigned(i), idx++});
        }
    }
    return out;
}

// ---- src/hpc.cpp:4-32 -------------------------------------------------------------------------
// run-collapse; the kept quality of a run is its maximum character.
void hpc(const std::string& seq, const std::string& qual, std::string& oseq, std::string& oqual)
{
    oseq.clear();
    oqual.clear();
    if (seq.empty()) return;
    char cb = seq[0], cq = qual[0];
    oseq += cb;
    for (size_t i = 1; i < seq.length(); i++) {
        if (seq[i] != cb) {
            cb = seq[i];
            oseq += cb;
            oqual += cq;
            cq = qual[i];
        } else if (cq < qual[i]) {
            cq = qual[i];
        }
    }
    oqual += cq;
}

// ---- src/util.cpp:13-38 ---------------------------------------------------------------------------
bool revcomp(const std::string& in, std::string& out)
{
    out.assign(in.rbegin(), in.rend());
    for (auto& c : out) {
        switch (c) {
            case 'A': c = 'T'; break;
            case 'C': c = 'G'; break;
            case 'G': c = 'C'; break;
            case 'T': c = 'A'; break;
            default: return false;  // the reference throws a std::string here
        }
    }
    return true;
}

// ---- src/util.cpp:6-10 -------------------------------------------------------------------------------
double round_dec(double x, int precision)
{
    int decimals = int(std::pow(10, precision));
    return (std::round(x * decimals)) / decimals;
}

// ---- src/qualscore.cpp:156-180 ----------------------------------------------------------------------------
std::vector<double> qual_tab(bool nomin)
{
    std::vector<double> t(129, 0.0);
    for (int i = 33; i <= 128; i++) {
        double v = pow(10, -((i - 33) / 10.0));
        if (!nomin && v > 0.79433) v = 0.79433;
        t[size_t(i)] = v;
    }
    return t;
}
const std::vector<double>& QT()
{
    static std::vector<double> t = qual_tab(false);
    return t;
}
const std::vector<double>& QTN()
{
    static std::vector<double> t = qual_tab(true);
    return t;
}

// ---- src/qualscore.cpp:107-136 ----------------------------------------------------------------------------
// expected number of error-free k-mers; running product updated as cur *= (p_enter / p_leave).
double qual_score(const std::string& q, int k, const std::vector<double>& tab)
{
    if (int(q.length()) <= k) return -1.0;
    std::deque<double> win;
    for (int i = 0; i < k; i++) win.push_back(1.0 - tab.at(size_t(int(q[size_t(i)]))));
    double cur = 1.0;
    for (auto& p : win) cur *= p;
    double sum = cur;
    for (size_t i = size_t(k); i < q.size(); i++) {
        double pe = 1.0 - tab.at(size_t(int(q[i])));
        double pl = win.front();
        win.pop_front();
        cur *= (pe / pl);
        sum += cur;
        win.push_back(pe);
    }
    return sum;
}

// ---- src/qualscore.cpp:147-154 ------------------------------------------------------------------------------
double error_rate(const std::string& q, const std::vector<double>& tab)
{
    double s = 0;
    for (auto c : q) s += tab.at(size_t(c));
    return s / double(q.length());
}

// ---- src/p_emp_prob.cpp:22-47 (table rows come from the extracted data file) ---------------------------------------
// rows with k == K and |w_row - W| <= 2, later rows overwrite earlier, symmetric in (e1,e2).
struct PTab {
    double p[15][15];
    int filled;
};
bool load_ptab(const char* path, int K, int W, PTab& t)
{
    for (auto& r : t.p)
        for (auto& c : r) c = std::nan("");
    t.filled = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    uint32_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "IOCPMIN1", 8) != 0 || fread(&n, 4, 1, f) != 1) {
        fclose(f);
        return false;
    }
    for (uint32_t c = 0; c < n; c++) {
        int32_t kw[2];
        double cells[225];
        if (fread(kw, 4, 2, f) != 2 || fread(cells, 8, 225, f) != 225) {
            fclose(f);
            return false;
        }
        if (kw[0] == K && std::abs(kw[1] - W) <= 2) {
            for (int a = 0; a < 15; a++)
                for (int b = 0; b < 15; b++)
                    if (!std::isnan(cells[a * 15 + b])) t.p[a][b] = cells[a * 15 + b];
        }
    }
    fclose(f);
    for (auto& r : t.p)
        for (auto& c : r)
            if (!std::isnan(c)) t.filled++;
    return true;
}

// ---- src/p_emp_prob.cpp:66-94 ------------------------------------------------------------------------------------------
// round each error rate to 2 decimals, clamp to [0.01, 0.15], exact-key lookup.  The keys of
// the reference map are the doubles parsed from "0.010000".."0.150000" == c/100 for c in 1..15,
// so the lookup is equality against c/100.
double pmin_lookup(const PTab& t, double e1, double e2, bool& ok)
{
    e1 = round_dec(e1, 2);
    e2 = round_dec(e2, 2);
    if (e1 > 0.15) e1 = 0.15;
    if (e1 < 0.01) e1 = 0.01;
    if (e2 > 0.15) e2 = 0.15;
    if (e2 < 0.01) e2 = 0.01;
    int a = -1, b = -1;
    for (int c = 1; c <= 15; c++) {
        if (double(c) / 100 == e1) a = c - 1;
        if (double(c) / 100 == e2) b = c - 1;
    }
    ok = (a >= 0 && b >= 0 && !std::isnan(t.p[a][b]));
    return ok ? t.p[a][b] : -1.0;  // reference: throws std::invalid_argument
}

// ---- data model: src/seq.h:20-98, src/cluster_data.h:14-29, src/serialize.h:23-103 ----------------------------
struct Seq {
    std::string name, seq, qual;
    double score = 0, err = 0;
};
struct ProcSeq {
    std::unique_ptr<Seq> Raw, Hpc;
    MzVec Mins, RevMins;
    int MatchStrand = 0;
    std::string Id;
    int orig = -1;   // oracle bookkeeping: index of the read in the original input
    bool repCopy = false;
};
typedef std::vector<std::shared_ptr<ProcSeq>> Cluster;
typedef std::vector<std::shared_ptr<Cluster>> Clusters;
struct Batch {
    int BatchNr = 0;
    unsigned long long BatchStart = 0, BatchEnd = 0, BatchBases = 0;
    int NrCls = 0;
    orc_params Args{};
    int Depth = 0;
    MinDB Db;
    Clusters Cls;
};
struct Reads {
    std::vector<std::unique_ptr<Seq>> v;
    std::vector<int> orig;
};

// ---- src/minimizer.cpp:31-42 --------------------------------------------------------------------------------------------------
void add_minimizers(const MzVec& mins, unsigned cls, MinDB& db, orc_stats* st)
{
    for (const auto& m : mins) {
        auto v = db.find(m.Min);
        if (v == db.end()) {
            db[m.Min] = std::vector<unsigned>{cls};
            if (st) st->index_appends++;
        } else if (v->second.size() == 0 || cls > v->second.back()) {
            v->second.emplace_back(cls);
            if (st) st->index_appends++;
        }
    }
}

// ---- src/minimizer.cpp:124-160: UpdateMinDB ------------------------------------------------------------------------------------
// After a consensus replaced the representative of cluster `best`: the cluster leaves the posting lists of the
// values only the old minimizers had (the list goes through a std::set: sorted, de-duplicated, `best` erased; a
// value without an entry gets an EMPTY one through operator[], and emptied lists are kept — the erase is
// commented out at :150-152) and is appended + sorted into the lists of the values only the new ones have.
void update_mindb(unsigned best, const MzVec& oldMins, const MzVec& newMins, MinDB& db)
{
    std::set<unsigned> oldSet, newSet, toIns, toDel;
    for (auto& m : oldMins) oldSet.insert(m.Min);
    for (auto& m : newMins) newSet.insert(m.Min);
    std::set_difference(oldSet.begin(), oldSet.end(), newSet.begin(), newSet.end(), std::inserter(toDel, toDel.begin()));
    std::set_difference(newSet.begin(), newSet.end(), oldSet.begin(), oldSet.end(), std::inserter(toIns, toIns.begin()));
    for (auto m : toDel) {
        auto& mins = db[m];
        std::set<unsigned> tmp(mins.begin(), mins.end());
        tmp.erase(best);
        mins.clear();
        mins.insert(mins.begin(), tmp.begin(), tmp.end());
    }
    for (auto m : toIns) {
        auto& tv = db[m];
        tv.push_back(best);
        std::sort(tv.begin(), tv.end());
    }
}

// ---- src/cluster.cpp:609-615 ---------------------------------------------------------------------------------------------------
void consolidate(const std::vector<RawHit>& raw, HitMap& res, int strand)
{
    for (auto& r : raw) res[std::make_pair(int(r.Cls), strand)].emplace_back(r.H);
}

// ---- src/minimizer.cpp:44-76 ---------------------------------------------------------------------------------------------------
HitMap minimizer_hits(const MzVec& mins, const MzVec& rev, const MinDB& db, orc_stats* st)
{
    std::vector<RawHit> raw;
    HitMap res(20 * (mins.size() + rev.size()), SClHash());
    raw.reserve(20 * mins.size());
    for (auto& m : mins) {
        auto it = db.find(m.Min);
        if (st) st->probes++;
        if (it != db.end())
            for (auto& c : it->second) raw.emplace_back(RawHit{c, Hit{m.Pos, m.Index}});
    }
    if (st) st->postings += raw.size();
    consolidate(raw, res, 1);
    raw.clear();
    for (auto& m : rev) {
        auto it = db.find(m.Min);
        if (st) st->probes++;
        if (it != db.end())
            for (auto& c : it->second) raw.emplace_back(RawHit{c, Hit{m.Pos, m.Index}});
    }
    if (st) st->postings += raw.size();
    consolidate(raw, res, -1);
    return res;
}

// ---- src/cluster.cpp:617-636 -----------------------------------------------------------------------------------------------------
bool by_size(const std::unique_ptr<SortedHit>& a, const std::unique_ptr<SortedHit>& b)
{
    return a->Size > b->Size;
}
SortedHits sort_hits(const HitMap& hits)
{
    SortedHits s;
    s.reserve(hits.size());
    for (auto& h : hits) {
        auto p = new SortedHit;
        p->Size = unsigned(h.second.size());
        p->Cls = unsigned(h.first.first);
        p->Strand = h.first.second;
        s.push_back(std::unique_ptr<SortedHit>(p));
    }
    std::sort(s.begin(), s.end(), by_size);
    return s;
}

// ---- tracing (test instrumentation, no counterpart in the reference) ----------------------------------------------------------
// SURVEY §8(c) golden item (3): the candidate table of chosen loop indices exactly as the reference's containers hold it
// when the loop reaches that entry — every (cls, strand) of the hit map with Size, the Index of its first hit, its
// position in the SortMinimizerHits order — and every getMappedRatio call (arguments -> totalMapped, length, ratio).
struct TraceRow {
    int entry, cls, strand;
    unsigned size, first_index, total_mapped;  // total_mapped: computed for EVERY candidate of a traced entry
    int order_pos;                             // position in the reference's SortedHits order
    int walked;                                // 1: the reference's walk really called getMappedRatio on it
};
struct MappedCall {
    int entry, cls, strand;
    unsigned total, hpc_len;
    double ratio, p_error;
};
std::set<int> g_trace_entries;
std::vector<TraceRow> g_trace_rows;
bool g_trace_mapped = false;
std::vector<MappedCall> g_mapped_calls;
thread_local int g_cur_entry = -1;
bool g_builtin_aligner = true;

// ---- src/cluster.cpp:324-353 -----------------------------------------------------------------------------------------------------
double mapped_ratio(const Seq& hpcSeq, const Seq& clHpc, const MzVec& mins, const HitVec& hits,
                    const PTab& tab, double minProbNoHits, bool& ok, double* pErrOut = nullptr, double* totalOut = nullptr)
{
    double pError = 1.0 - pmin_lookup(tab, clHpc.err, hpcSeq.err, ok);
    if (pErrOut) *pErrOut = pError;
    double total = 0;
    if (pow(pError, hits[0].Index) >= minProbNoHits) total += double(hits[0].Pos);
    for (unsigned i = 0; i < hits.size() - 1; i++) {
        auto& h1 = hits[i];
        auto& h2 = hits[i + 1];
        double np = pow(pError, double(h2.Index - (h1.Index + 1)));
        if (np >= minProbNoHits) total += double(h2.Pos - h1.Pos);
    }
    auto& h = hits[hits.size() - 1];
    if (pow(pError, double(mins.size() - (h.Index + 1))) >= minProbNoHits)
        total += hpcSeq.seq.length() - h.Pos;
    if (totalOut) *totalOut = total;
    return total / double(hpcSeq.seq.length());
}

const SCl NEG(-1, 0);

// ---- src/cluster.cpp:355-406 -----------------------------------------------------------------------------------------------------
SCl best_mapping(const ProcSeq& read, const Batch& left, const HitMap& hits, const SortedHits& order,
                 const PTab& tab, orc_stats* st, bool& ok)
{
    auto& args = left.Args;
    if (order.size() == 0) return NEG;
    unsigned top = order[0]->Size;
    if (top < unsigned(args.min_shared)) return NEG;
    SCl found = NEG;
    unsigned foundSize = 0;
    for (auto& c : order) {
        SCl scl(int(c->Cls), int(c->Strand));
        if (int(c->Size) < int(double(top) * args.min_fraction)) break;
        if (found.first >= 0) {
            // reference returns at the first pass; the oracle keeps walking only to COUNT
            // order-dependent ties (passing candidates of the same Size) for the statistics.
            if (c->Size < foundSize) break;
            bool ok2 = true;
            const MzVec& mm = (c->Strand == 1) ? read.Mins : read.RevMins;
            float mr2 = float(mapped_ratio(*read.Hpc, *(left.Cls.at(c->Cls)->at(0)->Hpc), mm,
                                           hits.at(scl), tab, args.min_prob_no_hits, ok2));
            if (mr2 >= args.mapped_threshold) {
                if (st) st->tie_reads++;
                break;
            }
            continue;
        }
        if (st) st->mapped_calls++;
        float mr;
        const MzVec& mm = (c->Strand == 1) ? read.Mins : read.RevMins;
        double pe = 0, tot = 0;
        const double mrd = mapped_ratio(*read.Hpc, *(left.Cls.at(c->Cls)->at(0)->Hpc), mm, hits.at(scl), tab,
                                        args.min_prob_no_hits, ok, &pe, &tot);
        mr = float(mrd);
        if (!ok) return NEG;
        if (g_trace_mapped)
            g_mapped_calls.push_back(MappedCall{g_cur_entry, int(c->Cls), int(c->Strand), unsigned(tot),
                                                unsigned(read.Hpc->seq.length()), mrd, pe});
        if (mr >= args.mapped_threshold) {
            found = scl;
            foundSize = c->Size;
            if (!st) break;
        }
    }
    return found;
}

// optional aligner hook for sahlin/furious (the reference calls parasail, cluster.cpp:461-515)
typedef int (*aligner_fn)(const char* read, int nread, const char* rep, int nrep, int gap_open,
                          int gap_extend, char* comp_out, int comp_cap);
aligner_fn g_aligner = nullptr;

// ---- src/cluster.cpp:425-440 -----------------------------------------------------------------------------------------------------
int gap_open_for(double e)
{
    if (e <= 0.01) return 5;
    if (e <= 0.04) return 4;
    if (e <= 0.1) return 3;
    return 2;
}

// ---- src/cluster.cpp:442-459 -----------------------------------------------------------------------------------------------------
double aln_ratio(const std::string& comp, double e, unsigned slen, unsigned k)
{
    double aligned = 0;
    auto limit = floor((1.0 - e) * k);
    if (comp.size() < k) return 0.0;  // reference: std::next past end (UB); unreachable for real reads
    auto i = comp.begin();
    auto j = std::next(i, k);
    while (j != comp.end()) {
        auto nm = std::count(i, j, '|');
        if (nm >= limit) aligned++;
        ++i;
        ++j;
    }
    return aligned / slen;
}

// ---- the call of src/cluster.cpp:408-423 + :498-503 (ParasailAlign + parasail_result_get_traceback) ----------------------------
// The oracle's own statement of parasail's semi-global alignment with traceback, from the library's published
// semantics (the source is not in /root/reference): rows = read, columns = representative; H(i,0) = H(0,j) = 0
// (leading end gaps free); E(i,j) = max(H(i,j-1) - open, E(i,j-1) - extend) is a gap in the read (a move along
// the representative), F(i,j) = max(H(i-1,j) - open, F(i-1,j) - extend) a gap in the representative;
// H = max(diagonal, E, F); the alignment ends at the best cell of the last column or last row (trailing end gaps
// free).  Where the published semantics leave a choice, this statement fixes it and the product has to follow:
// a tie between opening and extending a gap is an OPENING; H prefers the diagonal, then E, then F; the end cell is
// the first maximum met walking down the last column, then along the last row (strictly greater replaces).
// `comp` gets the bar character (0x7C) where two identical bases are paired and a blank everywhere else, end-gap columns
// included, like parasail_result_get_traceback(..., 0x7C, ' ', ' ').  Whole-matrix, one byte per cell: test-sized
// and baseline use only.
int sg_trace(const std::string& read, const std::string& rep, int match, int mismatch, int open, int extend,
             std::string& comp, int* score)
{
    const size_t n = read.size(), m = rep.size();
    const int NEGI = -(1 << 29);
    // per cell: bits 0-1 where H came from (0 diagonal, 1 E, 2 F), bit 2 E extended, bit 3 F extended
    std::vector<unsigned char> dir((n + 1) * (m + 1), 0);
    // Two sweeps per row.  Sweep A has no dependence along the row (the compiler vectorizes it): F and the diagonal
    // candidate of every cell from the previous row.  Sweep B carries E and H along the row.
    std::vector<int> hrow(m + 1, 0), frow(m + 1, NEGI), dg(m + 1, 0), fb(m + 1, 0), repi(m + 1, 0);
    for (size_t j = 0; j < m; j++) repi[j] = (unsigned char)rep[j];
    int bestv = NEGI;
    size_t bi = n, bj = m;
    for (size_t i = 1; i <= n; i++) {
        const int rc = (unsigned char)read[i - 1];
        int* __restrict h = hrow.data();
        int* __restrict f = frow.data();
        int* __restrict d = dg.data();
        int* __restrict fbit = fb.data();
        const int* __restrict rp = repi.data();
        const int mm = int(m);
        for (int j = 1; j <= mm; j++) {  // (h is only read here: it still holds the previous row)
            const int fo = h[j] - open, fe = f[j] - extend;
            fbit[j] = fe > fo ? 8 : 0;
            f[j] = fe > fo ? fe : fo;
            d[j] = h[j - 1] + (rc == rp[j - 1] ? match : mismatch);
        }
        h[0] = 0;
        int e = NEGI, hl = 0;  // hl = H(i, j-1)
        unsigned char* __restrict out = &dir[i * (m + 1)];
        for (size_t j = 1; j <= m; j++) {
            const int eo = hl - open, ee = e - extend;
            const unsigned char ebit = (unsigned char)(ee > eo ? 4 : 0);
            e = ee > eo ? ee : eo;
            const int dj = d[j], fj = f[j];
            const int mde = e > dj ? e : dj;  // (off the carried chain: only `from` needs it)
            const unsigned char from = (unsigned char)(fj > mde ? 2 : (e > dj ? 1 : 0));
            const int tdf = fj > dj ? fj : dj;
            hl = e > tdf ? e : tdf;           // carried chain: hl -> hl - open -> e -> hl
            h[j] = hl;
            out[j] = (unsigned char)(from | ebit | fbit[j]);
        }
        if (hrow[m] > bestv) {
            bestv = hrow[m];
            bi = i;
            bj = m;
        }
    }
    for (size_t j = 0; j <= m; j++) {
        const int v = n == 0 ? 0 : hrow[j];
        if (v > bestv) {
            bestv = v;
            bi = n;
            bj = j;
        }
    }
    if (score) *score = bestv;
    std::string back;
    back.reserve(n + m);
    back.append(m - bj, ' ');
    back.append(n - bi, ' ');
    size_t i = bi, j = bj;
    int in = 0;  // 0: in H, 1: inside an E gap, 2: inside an F gap
    while (i > 0 && j > 0) {
        const unsigned char b = dir[i * (m + 1) + j];
        if (in == 0) {
            const int from = b & 3;
            if (from == 0) {
                back.push_back(read[i - 1] == rep[j - 1] ? '|' : ' ');
                i--;
                j--;
            } else
                in = from;
        } else if (in == 1) {
            back.push_back(' ');
            if (!(b & 4)) in = 0;
            j--;
        } else {
            back.push_back(' ');
            if (!(b & 8)) in = 0;
            i--;
        }
    }
    back.append(j, ' ');
    back.append(i, ' ');
    comp.assign(back.rbegin(), back.rend());
    return int(comp.size());
}

// ---- src/cluster.cpp:461-515 -----------------------------------------------------------------------------------------------------
SCl best_aln(const ProcSeq& read, const SortedHits& order, const Batch& left, int& status)
{
    if (order.size() == 0) return NEG;
    if (!g_aligner && !g_builtin_aligner) {
        status = -2;
        return NEG;
    }
    unsigned top = order[0]->Size;
    const std::string& rs = read.Raw->seq;
    for (auto& c : order) {
        if (c->Size < top) break;
        auto& rep = left.Cls[c->Cls]->at(0)->Raw;
        std::string repSeq = rep->seq;
        if (c->Strand == -1) {
            std::string t;
            revcomp(repSeq, t);
            repSeq = t;
        }
        double e = read.Raw->err + rep->err;
        std::string compStr;
        if (g_aligner) {
            std::vector<char> comp(rs.size() + repSeq.size() + 2);
            int n = g_aligner(rs.c_str(), int(rs.size()), repSeq.c_str(), int(repSeq.size()), gap_open_for(e),
                              1, comp.data(), int(comp.size()));
            if (n < 0) {
                status = -3;
                return NEG;
            }
            compStr.assign(comp.data(), size_t(n));
        } else {
            sg_trace(rs, repSeq, 2, -2, gap_open_for(e), 1, compStr, nullptr);  // match 2, mismatch -2, extend 1: cluster.cpp:474-476
        }
        double r = aln_ratio(compStr, e, unsigned(rs.size()), unsigned(left.Args.k));
        if (r >= left.Args.aligned_threshold) return std::make_pair(int(c->Cls), c->Strand);
    }
    return NEG;
}

// ---- src/cluster.cpp:530-568 -----------------------------------------------------------------------------------------------------
SCl best_cluster(unsigned rightId, Batch& left, Batch& right, const PTab& tab, orc_stats* st, int& status)
{
    int mode = left.Args.mode;
    auto& read = right.Cls[rightId]->at(0);
    auto hits = minimizer_hits(read->Mins, read->RevMins, left.Db, st);
    auto order = sort_hits(hits);
    if (st) st->queries++;
    g_cur_entry = int(rightId);
    std::vector<size_t> traceAt;
    if (g_trace_entries.count(int(rightId))) {
        for (size_t pos = 0; pos < order.size(); pos++) {
            auto& c = order[pos];
            SCl scl(int(c->Cls), int(c->Strand));
            const HitVec& hv = hits.at(scl);
            bool ok2 = true;
            double tot = 0;
            const MzVec& mm = (c->Strand == 1) ? read->Mins : read->RevMins;
            mapped_ratio(*read->Hpc, *(left.Cls.at(c->Cls)->at(0)->Hpc), mm, hv, tab, left.Args.min_prob_no_hits, ok2, nullptr, &tot);
            traceAt.push_back(g_trace_rows.size());
            g_trace_rows.push_back(TraceRow{int(rightId), int(c->Cls), int(c->Strand), c->Size, hv[0].Index, unsigned(tot), int(pos), 0});
        }
    }
    const size_t callsBefore = g_mapped_calls.size();
    struct MarkWalked {  // on every way out: mark the traced rows the walk really evaluated
        const std::vector<size_t>& at;
        size_t before;
        bool on;
        ~MarkWalked()
        {
            if (!on) return;
            for (size_t x = before; x < g_mapped_calls.size(); x++)
                for (size_t r : at)
                    if (g_trace_rows[r].cls == g_mapped_calls[x].cls && g_trace_rows[r].strand == g_mapped_calls[x].strand) g_trace_rows[r].walked = 1;
        }
    } markWalked{traceAt, callsBefore, !traceAt.empty() && g_trace_mapped};
    if (order.size() == 0) return NEG;
    if (mode == 0 || mode == 1) {
        bool ok = true;
        auto m = best_mapping(*read, left, hits, order, tab, st, ok);
        if (!ok) {
            status = -4;
            return NEG;
        }
        if (m.first > -1) return m;
    }
    if (order[0]->Size < unsigned(left.Args.min_shared)) return NEG;
    if (mode == 1) return NEG;
    if (mode == 2 || mode == 0) {
        if (st) st->aln_invoked++;
        return best_aln(*read, order, left, status);
    }
    return NEG;
}

bool same_sort_args(const orc_params& a, const orc_params& b)
{
    // src/args.cpp:426-457 compares KmerSize, WindowSize, MinShared, MinQual, MappedThreshold,
    // AlignedThreshold, MinFraction, MinProbNoHits, Mode
    return a.k == b.k && a.w == b.w && a.min_shared == b.min_shared && a.min_qual == b.min_qual &&
           a.mapped_threshold == b.mapped_threshold && a.aligned_threshold == b.aligned_threshold &&
           a.min_fraction == b.min_fraction && a.min_prob_no_hits == b.min_prob_no_hits &&
           a.mode == b.mode;
}

// ---- consensus (src/consensus.cpp, src/cluster.cpp:200-204, 263-309) --------------------------------------------------------
// spoa is absent from /root/reference: the per-cluster partial-order graphs live behind a hook with exactly the
// operations the reference performs on them (create + seed, sequences().size(), AddSeqToGraph, GenerateConsensus,
// ConsPurge).  Everything around them — when a consensus is taken, the weighted error rates, the quirks of
// UpdateClusterConsensus, the re-minimizing of the new representative, UpdateMinDB — is restated here.
orc_cons_ops g_cons{};
bool g_cons_set = false;
int g_cons_min_size = 50, g_cons_period = 500;  // CmdArgs defaults, src/args.h:18-20

// src/consensus.cpp:34-126.  Returns 1 if the representative was replaced, 0 if not, < 0 on hook errors.
int update_cluster_consensus(const std::string& consName, Cluster& cl, int best, int rightEntry, const std::string& readSeq,
                             double readRawErr, double readHpcErr, int matchStrand, int consMinSize, int k, int w)
{
    const int leftSize = g_cons.size(g_cons.user, 0, best);                          // :41
    if (leftSize < 0) return -30;
    int rightSize = 1;                                                              // :42-43
    std::string rs = readSeq;
    if (matchStrand == -1) {
        std::string dropped;
        revcomp(rs, dropped);  // :47-49: RevComp(rs) returns a new string that is discarded — rs stays as it is
    }
    const int rsz = g_cons.size(g_cons.user, 1, rightEntry);
    const bool haveRight = rsz >= 0;
    if (haveRight) rightSize = rsz;                                                 // :51-53
    ProcSeq* clsRep = cl.at(0).get();
    const double hpcErr = (clsRep->Hpc->err * double(leftSize) + readHpcErr * double(rightSize)) / double(leftSize + rightSize);
    const double rawErr = (clsRep->Raw->err * double(leftSize) + readRawErr * double(rightSize)) / double(leftSize + rightSize);
    if (g_cons.add(g_cons.user, 0, best, rs.data(), int(rs.size()), haveRight ? unsigned(rightSize) : 1u) < 0) return -31;  // :76-81
    if (g_cons.size(g_cons.user, 0, best) < consMinSize) return 0;                  // :83-85
    std::string cons;
    {
        std::vector<char> buf(size_t(1) << 22);
        int n = g_cons.consensus(g_cons.user, 0, best, buf.data(), int(buf.size()));
        if (n < 0) return -32;
        cons.assign(buf.data(), size_t(n));
    }
    auto& rep = cl[0];
    rep->Raw->seq = cons;                                                           // :93-97
    rep->Raw->name = consName;
    rep->Raw->err = rawErr;
    rep->Raw->score = rawErr * double(cons.length());
    const char fixedQualRaw = std::to_string(int(-10 * log10(rawErr)) + 33)[0];     // :98-99: first CHARACTER of the number
    rep->Raw->qual = std::string(cons.length(), fixedQualRaw);
    std::unique_ptr<Seq> hpcSeq(new Seq);
    if (cons.length() > unsigned(2 * k) || cons.length() >= unsigned(w)) {          // :104-118
        hpcSeq->name = rep->Raw->name;
        hpc(rep->Raw->seq, rep->Raw->qual, hpcSeq->seq, hpcSeq->qual);
        hpcSeq->err = hpcErr;
        hpcSeq->score = hpcErr * double(hpcSeq->seq.length());
        if (hpcSeq->seq.length() < unsigned(2 * k) || hpcSeq->seq.length() < unsigned(w)) {
            hpcSeq->score = -1.0;
            rep->Raw->score = -1.0;
            rep->Raw->err = 0.9999;
            hpcSeq->err = 0.9999;
        }
    }
    if (hpcSeq->seq.length() <= unsigned(w - k)) return -33;  // GetKmerMinimizers would read past its input
    auto kmerSeq = kmer_encode(hpcSeq->seq, unsigned(k));                            // :119-120
    std::string rc;
    if (!revcomp(hpcSeq->seq, rc)) return -34;
    auto revKmerSeq = kmer_encode(rc, unsigned(k));
    hpcSeq->err = hpcErr;                                                           // :121
    rep->Hpc = std::move(hpcSeq);
    rep->Mins = minimizers(kmerSeq, k, w);                                          // :123-124
    rep->RevMins = minimizers(revKmerSeq, k, w);
    return 1;
}

// ---- src/cluster.cpp:67-322 ------------------------------------------------------------------------------------------------
int cluster_sorted_reads(Batch& left, Batch& right, const char* binpath, orc_stats* st)
{
    if (!same_sort_args(left.Args, right.Args)) return -10;
    auto args = left.Args;
    if (right.Depth > 0 && right.BatchStart != (left.BatchEnd + 1)) return -11;
    if (left.Depth > 0 && right.Depth > left.Depth) return -12;
    if (left.Db.size() == 0) left.Db = MinDB(1000000, IdHash());
    right.Db = MinDB(0, IdHash());
    const int consMaxSize = args.cons_max_size;
    if (consMaxSize > 0 && !g_cons_set) return -13;  // consensus needs the graph hook (orc_set_consensus)
    auto& cls = left.Cls;
    auto& reads = right.Cls;
    PTab tab;
    if (!load_ptab(binpath, args.k, args.w, tab)) return -14;
    int minCls = left.Args.min_cls_size;

    for (unsigned i = 0; i < reads.size(); i++) {
        if (reads[i] == nullptr || reads[i]->size() == 0) continue;
        if ((right.Depth > 0) && (minCls > 1) && (int(reads[i]->size() - 1) < minCls)) continue;
        auto& read = reads[i]->at(0);
        if (read == nullptr || read->Raw == nullptr) continue;
        auto& seq = read->Raw;
        if (read->Hpc == nullptr) return -15;  // reference dereferences a null HpcSeq here (:129-130)
        if (seq->score < 0) continue;
        if (seq->seq.length() < unsigned(2 * args.k)) {
            seq->score = -1.0;
            continue;
        }
        if (read->Hpc->seq.length() < unsigned(2 * args.k)) {
            seq->score = -1.0;
            continue;
        }
        if ((-10 * log10(seq->err)) <= args.min_qual) {
            seq->score = -1.0;
            continue;
        }
        int status = 0;
        SCl st_match = best_cluster(i, left, right, tab, st, status);
        if (status < 0) return status;
        int best = st_match.first;

        if (best == -1) {
            unsigned newId = unsigned(cls.size());
            auto nr = reads[i]->size();
            add_minimizers(read->Mins, newId, left.Db, st);
            if (nr == 1) {
                auto rep = reads[i]->at(0);
                auto n = std::make_shared<ProcSeq>();
                n->Raw.reset(new Seq(*rep->Raw));
                n->Hpc.reset(new Seq(*rep->Hpc));
                n->Mins = rep->Mins;
                n->RevMins = rep->RevMins;
                n->MatchStrand = rep->MatchStrand;
                n->Id = rep->Id;
                n->orig = rep->orig;
                n->repCopy = true;
                std::string nm = "rep_" + std::to_string(left.BatchNr) + "_" + std::to_string(newId);
                n->Raw->name = nm;
                n->Hpc->name = nm;
                reads[i]->insert(reads[i]->begin(), 1, n);
            }
            if (g_cons_set) {  // cluster.cpp:200-204: a graph seeded with the representative, whatever ConsMaxSize is
                const std::string& rseq = reads[i]->at(0)->Raw->seq;
                if (g_cons.create(g_cons.user, 0, int(newId), rseq.data(), int(rseq.size())) < 0) return -35;
            }
            cls.emplace_back(reads[i]);
            if (nr == 1 && cls[newId]->size() != 2) return -16;
            left.NrCls++;
            if (st) st->new_clusters++;
        } else {
            auto startIt = reads[i]->begin();
            const std::string readSeq = (*startIt)->Raw->seq;  // cluster.cpp:172-175: taken before the members are cleared
            const double readRawErr = (*startIt)->Raw->err, readHpcErr = (*startIt)->Hpc->err;
            for (unsigned j = 0; j < reads[i]->size(); j++) {
                auto& s = reads[i]->at(j);
                if (s == nullptr) return -17;
                if (st_match.second == -1) {
                    if (s->MatchStrand == 1)
                        s->MatchStrand = -1;
                    else if (s->MatchStrand == -1)
                        s->MatchStrand = 1;
                    else
                        return -18;
                }
                s->Mins = MzVec(0);
                s->RevMins = MzVec(0);
            }
            if (reads[i]->size() > 1) startIt++;
            std::move(startIt, std::end(*(reads[i])), std::back_inserter(*(cls[size_t(best)])));
            if (st) st->joins++;
            // ---- consensus branch, cluster.cpp:263-309 ----
            if (consMaxSize <= 0) continue;
            if ((left.Depth == -1) && (g_cons_period > 0) && (int(cls[size_t(best)]->size()) > g_cons_period)) continue;
            const std::string consName = "cons_" + std::to_string(left.BatchNr) + "_" + std::to_string(i);
            const MzVec oldMins = cls.at(size_t(best))->at(0)->Mins;
            int consMinSize = g_cons_min_size;
            if (left.Depth != -1) consMinSize = 2;  // FIXME of the reference, :286-288
            const int ok = update_cluster_consensus(consName, *(cls[size_t(best)]), best, int(i), readSeq, readRawErr, readHpcErr,
                                                    st_match.second, consMinSize, args.k, args.w);
            if (ok < 0) return ok;
            if (ok) {
                if (st) st->cons_invoked++;
                update_mindb(unsigned(best), oldMins, cls[size_t(best)]->at(0)->Mins, left.Db);
            }
            if (ok && g_cons.size(g_cons.user, 0, best) > consMaxSize) {  // ConsPurge, consensus.cpp:128-137
                const std::string& repSeq = cls[size_t(best)]->at(0)->Raw->seq;
                const int wgt = g_cons.size(g_cons.user, 0, best);
                if (g_cons.purge(g_cons.user, 0, best, repSeq.data(), int(repSeq.size()), unsigned(wgt)) < 0) return -36;
            }
        }
    }
    left.Depth++;
    left.BatchEnd = right.BatchEnd;
    left.BatchBases = left.BatchBases + right.BatchBases;
    return 0;
}

}  // namespace

// =====================================================================================================
extern "C" {

void orc_set_aligner(void* fn) { g_aligner = reinterpret_cast<aligner_fn>(fn); }
void orc_use_builtin_aligner(int on) { g_builtin_aligner = on != 0; }

int orc_align(const char* read, int nread, const char* rep, int nrep, int match, int mismatch, int gap_open, int gap_extend,
              char* comp, int comp_cap, int* score)
{
    std::string c;
    const int n = sg_trace(std::string(read, size_t(nread)), std::string(rep, size_t(nrep)), match, mismatch, gap_open, gap_extend, c, score);
    if (comp) {
        if (n + 1 > comp_cap) return -1;
        memcpy(comp, c.data(), size_t(n));
        comp[n] = 0;
    }
    return n;
}
int orc_gap_open(double e) { return gap_open_for(e); }
double orc_aln_ratio(const char* comp, int n, double e, unsigned slen, unsigned k) { return aln_ratio(std::string(comp, size_t(n)), e, slen, k); }

void orc_trace_set(const int32_t* entries, int n, int mapped_calls)
{
    g_trace_entries.clear();
    for (int i = 0; i < n; i++) g_trace_entries.insert(entries[i]);
    g_trace_mapped = mapped_calls != 0 || n > 0;
    g_trace_rows.clear();
    g_mapped_calls.clear();
}
int64_t orc_trace_rows(int32_t* entry, int32_t* cls, int32_t* strand, uint32_t* size, uint32_t* first_index, uint32_t* total_mapped,
                       int32_t* order_pos, uint8_t* walked)
{
    if (entry)
        for (size_t i = 0; i < g_trace_rows.size(); i++) {
            const TraceRow& r = g_trace_rows[i];
            entry[i] = r.entry;
            cls[i] = r.cls;
            strand[i] = r.strand;
            size[i] = r.size;
            first_index[i] = r.first_index;
            total_mapped[i] = r.total_mapped;
            order_pos[i] = r.order_pos;
            walked[i] = uint8_t(r.walked);
        }
    return int64_t(g_trace_rows.size());
}
int64_t orc_trace_mapped_calls(int32_t* entry, int32_t* cls, int32_t* strand, uint32_t* total, uint32_t* hpc_len, double* ratio, double* p_error)
{
    if (entry)
        for (size_t i = 0; i < g_mapped_calls.size(); i++) {
            const MappedCall& r = g_mapped_calls[i];
            entry[i] = r.entry;
            cls[i] = r.cls;
            strand[i] = r.strand;
            total[i] = r.total;
            hpc_len[i] = r.hpc_len;
            ratio[i] = r.ratio;
            p_error[i] = r.p_error;
        }
    return int64_t(g_mapped_calls.size());
}

void orc_set_consensus(const orc_cons_ops* ops, int cons_min_size, int cons_period)
{
    g_cons_set = ops != nullptr;
    if (ops) g_cons = *ops;
    g_cons_min_size = cons_min_size;
    g_cons_period = cons_period;
}

int orc_hpc(const char* seq, const char* qual, int n, char* oseq, char* oqual)
{
    std::string a, b;
    hpc(std::string(seq, size_t(n)), std::string(qual, size_t(n)), a, b);
    memcpy(oseq, a.data(), a.size());
    memcpy(oqual, b.data(), b.size());
    return int(a.size());
}

int orc_revcomp(const char* seq, int n, char* out)
{
    std::string o;
    if (!revcomp(std::string(seq, size_t(n)), o)) return -1;
    memcpy(out, o.data(), o.size());
    return 0;
}

int orc_kmer_encode(const char* seq, int n, int k, uint32_t* out)
{
    auto v = kmer_encode(std::string(seq, size_t(n)), unsigned(k));
    if (out && !v.empty()) memcpy(out, v.data(), v.size() * 4);  // (an empty vector's data() may be null: memcpy(_, null, 0) is UB)
    return int(v.size());
}

int orc_minimizers(const uint32_t* kmers, int n, int k, int w, uint32_t* omin, uint32_t* opos, uint32_t* oidx)
{
    std::vector<unsigned> ks(kmers, kmers + n);
    auto m = minimizers(ks, k, w);
    for (size_t i = 0; i < m.size(); i++) {
        if (omin) omin[i] = m[i].Min;
        if (opos) opos[i] = m[i].Pos;
        if (oidx) oidx[i] = m[i].Index;
    }
    return int(m.size());
}

void orc_qual_tab(int nomin, double* out129)
{
    auto t = qual_tab(nomin != 0);
    memcpy(out129, t.data(), 129 * sizeof(double));
}

double orc_qual_score(const char* qual, int n, int k) { return qual_score(std::string(qual, size_t(n)), k, QT()); }

double orc_error_rate(const char* qual, int n, int nomin)
{
    return error_rate(std::string(qual, size_t(n)), nomin ? QTN() : QT());
}

double orc_round(double x, int precision) { return round_dec(x, precision); }

int orc_pmin_table(const char* binpath, int k, int w, double* out225)
{
    PTab t;
    if (!load_ptab(binpath, k, w, t)) return -1;
    memcpy(out225, t.p, sizeof(t.p));
    return t.filled;
}

double orc_pmin_lookup(const double* tab225, double e1, double e2, int* err)
{
    PTab t;
    memcpy(t.p, tab225, sizeof(t.p));
    bool ok = true;
    double r = pmin_lookup(t, e1, e2, ok);
    if (err) *err = ok ? 0 : -1;
    return r;
}

// largest n with pow(1 - p, n) >= minProbNoHits (the predicate of cluster.cpp:333-347)
int orc_gap_limit(double p_shared, double min_prob_no_hits)
{
    double pe = 1.0 - p_shared;
    int n = 0;
    while (n < 1000000 && pow(pe, double(n + 1)) >= min_prob_no_hits) n++;
    if (!(pow(pe, 0.0) >= min_prob_no_hits)) return -1;
    return n;
}

uint32_t orc_kmer_to_index(const char* kmer, int k)
{
    unsigned v = 0;
    for (int j = 0; j < k; j++) v = 4u * v + base_code(kmer[j]);
    return v;
}

// src/kmer_index.h:47-57
void orc_index_to_kmer(uint32_t idx, int k, char* out)
{
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (int j = k - 1; j >= 0; j--) {
        out[j] = L[idx % 4];
        idx /= 4;
    }
}

int orc_minmatch(const char* ref, const char* refq, int nref, const char* read, const char* readq, int nread,
                 int k, int w, const char* binpath, double min_prob_no_hits, uint32_t* top_size,
                 double* p_error, double* mapped)
{
    Seq R, Q;
    hpc(std::string(ref, size_t(nref)), std::string(refq, size_t(nref)), R.seq, R.qual);
    hpc(std::string(read, size_t(nread)), std::string(readq, size_t(nread)), Q.seq, Q.qual);
    MinDB db;
    auto rm = minimizers(kmer_encode(R.seq, unsigned(k)), k, w);
    add_minimizers(rm, 1, db, nullptr);
    auto qm = minimizers(kmer_encode(Q.seq, unsigned(k)), k, w);
    auto hits = minimizer_hits(qm, MzVec(), db, nullptr);
    auto order = sort_hits(hits);
    if (order.empty()) return -1;
    *top_size = order[0]->Size;
    PTab tab;
    if (!load_ptab(binpath, k, w, tab)) return -2;
    Q.err = error_rate(Q.qual, QT());  // the reference test uses InitQualTab() here (test :124-126)
    R.err = error_rate(R.qual, QT());
    bool ok = true;
    *mapped = mapped_ratio(Q, R, qm, hits[std::make_pair(1, 1)], tab, min_prob_no_hits, ok, p_error);
    return ok ? 0 : -3;
}

// ---- reads ---------------------------------------------------------------------------------------------
void* orc_reads_new(const char* seqs, const char* quals, const int64_t* offs, int n)
{
    auto r = new Reads;
    for (int i = 0; i < n; i++) {
        auto s = std::unique_ptr<Seq>(new Seq);
        s->name = "r" + std::to_string(i);
        s->seq.assign(seqs + offs[i], size_t(offs[i + 1] - offs[i]));
        s->qual.assign(quals + offs[i], size_t(offs[i + 1] - offs[i]));
        r->v.push_back(std::move(s));
        r->orig.push_back(i);
    }
    return r;
}
void orc_reads_free(void* h) { delete static_cast<Reads*>(h); }

// src/qualscore.cpp:14-37 + :138-145
void orc_reads_score_sort(void* h, int k, int /*w*/)
{
    auto r = static_cast<Reads*>(h);
    for (auto& s : r->v) {
        if (s->seq.length() > unsigned(2 * k)) {
            double qs = qual_score(s->qual, k, QT());
            if (qs <= 0) qs = -1.0;
            s->score = qs;
            s->err = error_rate(s->qual, QTN());
        } else {
            s->score = -1.0;
            s->err = 1.0;
        }
    }
    // stable sort of (seq, orig) pairs, descending score
    std::vector<size_t> perm(r->v.size());
    for (size_t i = 0; i < perm.size(); i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(),
                     [&](size_t a, size_t b) { return r->v[a]->score > r->v[b]->score; });
    std::vector<std::unique_ptr<Seq>> nv;
    std::vector<int> no;
    for (auto p : perm) {
        nv.push_back(std::move(r->v[p]));
        no.push_back(r->orig[p]);
    }
    r->v.swap(nv);
    r->orig.swap(no);
}
int orc_reads_n(void* h) { return int(static_cast<Reads*>(h)->v.size()); }
// test plumbing: shift the original-read ids of a read set (several independently sorted read sets folded into one
// clustering keep distinct ids)
void orc_reads_shift_orig(void* h, int base)
{
    for (auto& o : static_cast<Reads*>(h)->orig) o += base;
}
void orc_reads_order(void* h, int32_t* orig, double* score, double* err)
{
    auto r = static_cast<Reads*>(h);
    for (size_t i = 0; i < r->v.size(); i++) {
        if (orig) orig[i] = r->orig[i];
        if (score) score[i] = r->v[i]->score;
        if (err) err[i] = r->v[i]->err;
    }
}

// ---- src/qualscore.cpp:39-105 ------------------------------------------------------------------------------
void* orc_batch_prepare(void* reads, int start, int end, const orc_params* p, int batch_nr)
{
    auto r = static_cast<Reads*>(reads);
    auto b = new Batch;
    int size = 1 + end - start;
    b->Cls = Clusters(size_t(size));
    int k = p->k, w = p->w;
    unsigned long long bases = 0;
    for (int i = 0; i < size; i++) {
        auto& s = r->v[size_t(start + i)];
        int orig = r->orig[size_t(start + i)];
        bases += s->seq.length();
        b->Cls[size_t(i)] = std::make_shared<Cluster>();
        auto ps = std::make_shared<ProcSeq>();
        ps->Id = s->name;
        ps->orig = orig;
        if ((-10 * log10(s->err)) <= p->min_qual) {
            b->Cls[size_t(i)]->push_back(ps);  // {nullptr, nullptr, {}, {}, 0, name}
            continue;
        }
        if (s->seq.length() > unsigned(2 * k) || s->seq.length() >= unsigned(w)) {
            std::unique_ptr<Seq> h(new Seq);
            h->name = s->name;
            h->score = s->score;
            hpc(s->seq, s->qual, h->seq, h->qual);
            if (h->seq.length() < unsigned(2 * k) || h->seq.length() < unsigned(w)) {
                s->score = -1.0;
                b->Cls[size_t(i)]->push_back(ps);
                continue;
            }
            auto ks = kmer_encode(h->seq, unsigned(k));
            std::string rc;
            if (!revcomp(h->seq, rc)) {  // reference: uncaught throw -> terminate
                delete b;
                return nullptr;
            }
            auto rks = kmer_encode(rc, unsigned(k));
            h->err = error_rate(h->qual, QTN());
            ps->Mins = minimizers(ks, k, w);
            ps->RevMins = minimizers(rks, k, w);
            ps->Raw.reset(new Seq(*s));
            ps->Hpc = std::move(h);
            ps->MatchStrand = 1;
            b->Cls[size_t(i)]->push_back(ps);
        } else {
            // reference: ProcSeq{move(s), nullptr, ...} with a use-after-move of s->Name(); such
            // reads (len <= 2k and len < w) crash the reference.  The oracle keeps the entry as a
            // raw-only record and orc_cluster reports -15 if it is ever reached.
            s->score = -1.0;
            ps->Raw.reset(new Seq(*s));
            b->Cls[size_t(i)]->push_back(ps);
        }
    }
    b->NrCls = int(b->Cls.size());
    b->BatchStart = (unsigned long long)start;
    b->BatchEnd = (unsigned long long)end;
    b->Depth = -1;
    b->BatchNr = batch_nr;
    b->BatchBases = bases;
    b->Args = *p;
    return b;
}
void orc_batch_free(void* b) { delete static_cast<Batch*>(b); }
int orc_batch_n_entries(void* b) { return int(static_cast<Batch*>(b)->Cls.size()); }

void orc_batch_entry_info(void* bh, int32_t* state, int32_t* orig, int32_t* raw_len, int32_t* hpc_len,
                          double* score, double* raw_err, double* hpc_err, int32_t* n_fwd, int32_t* n_rev)
{
    auto b = static_cast<Batch*>(bh);
    for (size_t i = 0; i < b->Cls.size(); i++) {
        auto& e = b->Cls[i]->at(0);
        bool nul = (e->Raw == nullptr);
        if (state) state[i] = nul ? 1 : (e->Hpc == nullptr ? 2 : 0);
        if (orig) orig[i] = e->orig;
        if (raw_len) raw_len[i] = nul ? 0 : int(e->Raw->seq.size());
        if (hpc_len) hpc_len[i] = (e->Hpc == nullptr) ? 0 : int(e->Hpc->seq.size());
        if (score) score[i] = nul ? -1.0 : e->Raw->score;
        if (raw_err) raw_err[i] = nul ? 1.0 : e->Raw->err;
        if (hpc_err) hpc_err[i] = (e->Hpc == nullptr) ? 1.0 : e->Hpc->err;
        if (n_fwd) n_fwd[i] = int(e->Mins.size());
        if (n_rev) n_rev[i] = int(e->RevMins.size());
    }
}

int orc_batch_entry_mins(void* bh, int i, int strand, uint32_t* omin, uint32_t* opos, uint32_t* oidx)
{
    auto b = static_cast<Batch*>(bh);
    auto& e = b->Cls[size_t(i)]->at(0);
    const MzVec& m = strand == 0 ? e->Mins : e->RevMins;
    for (size_t j = 0; j < m.size(); j++) {
        if (omin) omin[j] = m[j].Min;
        if (opos) opos[j] = m[j].Pos;
        if (oidx) oidx[j] = m[j].Index;
    }
    return int(m.size());
}

int orc_batch_entry_hpc(void* bh, int i, char* oseq, char* oqual)
{
    auto b = static_cast<Batch*>(bh);
    auto& e = b->Cls[size_t(i)]->at(0);
    if (e->Hpc == nullptr) return 0;
    if (oseq) memcpy(oseq, e->Hpc->seq.data(), e->Hpc->seq.size());
    if (oqual) memcpy(oqual, e->Hpc->qual.data(), e->Hpc->qual.size());
    return int(e->Hpc->seq.size());
}

// ---- src/main.cpp:238-382 (the parts that touch the batches) + src/serialize.cpp:29-43 ------------------------
int orc_cluster(void* lh, void* rh, const orc_params* p, const char* binpath, orc_stats* st)
{
    auto left = static_cast<Batch*>(lh);
    std::unique_ptr<Batch> pseudo;
    Batch* right = static_cast<Batch*>(rh);
    if (st) memset(st, 0, sizeof(*st));
    if (right == nullptr) {
        pseudo.reset(new Batch);
        pseudo->BatchNr = -left->BatchNr;
        pseudo->BatchStart = left->BatchStart;
        pseudo->BatchEnd = left->BatchEnd;
        pseudo->BatchBases = 0;
        pseudo->Args = left->Args;
        pseudo->Depth = -1;
        pseudo->Cls = Clusters(left->Cls);
        pseudo->NrCls = int(pseudo->Cls.size());
        right = pseudo.get();
        left->Cls.clear();
        if (left->Depth > 0) left->Depth = -left->Depth;
        left->NrCls = 0;
        left->Db = MinDB(1000000, IdHash());
    } else {
        right->Db = MinDB(0, IdHash());
    }
    left->Args.mode = p->mode;
    right->Args.mode = p->mode;
    if (p->min_cls_size > 0) left->Args.min_cls_size = p->min_cls_size;
    return cluster_sorted_reads(*left, *right, binpath, st);
}

int orc_batch_n_clusters(void* b) { return int(static_cast<Batch*>(b)->Cls.size()); }
int orc_batch_n_members(void* bh)
{
    auto b = static_cast<Batch*>(bh);
    size_t n = 0;
    for (auto& c : b->Cls) n += c->size();
    return int(n);
}
int orc_batch_members(void* bh, int32_t* cls, int32_t* orig, int32_t* strand, int32_t* is_rep)
{
    auto b = static_cast<Batch*>(bh);
    size_t n = 0;
    for (size_t c = 0; c < b->Cls.size(); c++)
        for (auto& m : *b->Cls[c]) {
            cls[n] = int(c);
            orig[n] = m->orig;
            strand[n] = m->MatchStrand;
            is_rep[n] = m->repCopy ? 1 : 0;
            n++;
        }
    return int(n);
}

int64_t orc_batch_index(void* bh, uint32_t* keys, int64_t* offs, uint32_t* postings, int64_t* n_postings)
{
    auto b = static_cast<Batch*>(bh);
    std::vector<unsigned> ks;
    ks.reserve(b->Db.size());
    int64_t np = 0;
    for (auto& kv : b->Db) {
        ks.push_back(kv.first);
        np += int64_t(kv.second.size());
    }
    std::sort(ks.begin(), ks.end());
    if (n_postings) *n_postings = np;
    if (keys && offs && postings) {
        int64_t o = 0;
        for (size_t i = 0; i < ks.size(); i++) {
            keys[i] = ks[i];
            offs[i] = o;
            auto& v = b->Db[ks[i]];
            for (auto c : v) postings[o++] = c;
        }
        offs[ks.size()] = o;
    }
    return int64_t(ks.size());
}

// UpdateMinDB on a clustered batch's index; old / new minimizers are given by value only (Pos and Index play
// no part, src/minimizer.cpp:132-137).  With mins_too != 0 the representative's forward minimizers are
// replaced as well (values, Pos = Index = ordinal), so that a later update sees them as its old ones.
int orc_batch_update_mindb(void* bh, int cls, const uint32_t* old_min, int64_t n_old, const uint32_t* new_min,
                           int64_t n_new, int mins_too)
{
    auto b = static_cast<Batch*>(bh);
    if (!b || cls < 0 || size_t(cls) >= b->Cls.size()) return -1;
    MzVec o, nw;
    for (int64_t i = 0; i < n_old; ++i) o.push_back(Mz{old_min[i], unsigned(i), unsigned(i)});
    for (int64_t i = 0; i < n_new; ++i) nw.push_back(Mz{new_min[i], unsigned(i), unsigned(i)});
    update_mindb(unsigned(cls), o, nw, b->Db);
    if (mins_too && b->Cls[size_t(cls)] && !b->Cls[size_t(cls)]->empty() && b->Cls[size_t(cls)]->at(0)) b->Cls[size_t(cls)]->at(0)->Mins = nw;
    return 0;
}

}  // extern "C"
