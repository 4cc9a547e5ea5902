// ioc_align.cpp — host semi-global aligner for the sahlin / furious fallback.
//
// The reference calls parasail (src/cluster.cpp:408-423, 461-515): parasail_sg_trace_scan_16 / _32 with
// match 2, mismatch -2, gap open from setGapOpen(e1+e2) (2..5), gap extend 1, then
// parasail_result_get_traceback(..., '|', ' ', ' ') and getAlnRatio over the `comp` string.  parasail is
// a third-party library absent from /root/reference (.gitmodules:4-6, version unrecoverable): this is a
// from-scratch Gotoh aligner with the published semantics of that call (all four sequence ends free;
// a gap of length n costs open + (n-1)*extend; traceback covers the end-gap columns).  Parity with
// parasail's tie-breaking is UNPINNED except for the reference's single AlnRatioTest vector
// (test/isONclust2_test.cpp:137-181), which tests/test_align_host.py checks.
// The fallback stays on the host by design (BASELINE.json north_star).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "isonclust2_hip.h"

namespace {

// direction bits of the traceback matrix, 1 byte per cell
enum : uint8_t { H_DIAG = 0, H_FROM_E = 1, H_FROM_F = 2, H_MASK = 3, E_EXT = 4, F_EXT = 8 };

}  // namespace

extern "C" {

// Semi-global alignment of query (rows) against ref (columns).  Writes the comparison string of the
// whole alignment ('|' identical bases, ' ' otherwise, end-gap columns included) into comp (capacity
// comp_cap >= qlen + rlen + 1) and returns its length, or a negative ioc_status.
int ioc_host_align(const char* query, int32_t qlen, const char* ref, int32_t rlen, int32_t match,
                   int32_t mismatch, int32_t gap_open, int32_t gap_extend, char* comp, int32_t comp_cap,
                   int32_t* score_out)
{
    if (!query || !ref || !comp || qlen < 0 || rlen < 0) return IOC_ERR_ARG;
    if (comp_cap < qlen + rlen + 1) return IOC_ERR_CAPACITY;
    const int n = qlen, m = rlen;
    if (uint64_t(n + 1) * uint64_t(m + 1) > (1ull << 33)) return IOC_ERR_CAPACITY;
    const int NEG = INT32_MIN / 4;
    std::vector<uint8_t> tb(size_t(n + 1) * size_t(m + 1), 0);
    std::vector<int> H(size_t(m) + 1, 0), F(size_t(m) + 1, NEG), Hprev(size_t(m) + 1, 0);
    // free leading gaps on both sequences: first row and first column are 0
    int best = NEG, bi = n, bj = m;
    for (int i = 1; i <= n; ++i) {
        std::swap(H, Hprev);
        H[0] = 0;
        int E = NEG;
        const char qc = query[i - 1];
        uint8_t* row = tb.data() + size_t(i) * size_t(m + 1);
        for (int j = 1; j <= m; ++j) {
            // E: gap in the query (horizontal move), F: gap in the reference (vertical move)
            const int e_open = H[j - 1] - gap_open, e_ext = E - gap_extend;
            uint8_t d = 0;
            if (e_ext > e_open) {
                E = e_ext;
                d |= E_EXT;
            } else {
                E = e_open;
            }
            const int f_open = Hprev[j] - gap_open, f_ext = F[j] - gap_extend;
            if (f_ext > f_open) {
                F[j] = f_ext;
                d |= F_EXT;
            } else {
                F[j] = f_open;
            }
            const int diag = Hprev[j - 1] + (qc == ref[j - 1] ? match : mismatch);
            int h = diag;
            uint8_t from = H_DIAG;
            if (E > h) {
                h = E;
                from = H_FROM_E;
            }
            if (F[j] > h) {
                h = F[j];
                from = H_FROM_F;
            }
            H[j] = h;
            row[j] = uint8_t(d | from);
        }
        // free trailing gap on the reference: best of the last column
        if (H[m] > best) {
            best = H[m];
            bi = i;
            bj = m;
        }
    }
    // free trailing gap on the query: best of the last row
    if (n == 0) std::fill(H.begin(), H.end(), 0);
    for (int j = 0; j <= m; ++j) {
        const int v = (n == 0) ? 0 : H[j];
        if (v > best) {
            best = v;
            bi = n;
            bj = j;
        }
    }
    if (score_out) *score_out = best;
    // traceback from (bi, bj); trailing end gaps first (they are the tail of the strings)
    std::string rev;
    rev.reserve(size_t(n + m));
    for (int j = m; j > bj; --j) rev.push_back(' ');
    for (int i = n; i > bi; --i) rev.push_back(' ');
    int i = bi, j = bj, state = 0;  // 0 = H, 1 = E, 2 = F
    while (i > 0 && j > 0) {
        const uint8_t t = tb[size_t(i) * size_t(m + 1) + size_t(j)];
        if (state == 0) {
            const uint8_t from = t & H_MASK;
            if (from == H_DIAG) {
                rev.push_back(query[i - 1] == ref[j - 1] ? '|' : ' ');
                --i;
                --j;
            } else if (from == H_FROM_E) {
                state = 1;
            } else {
                state = 2;
            }
        } else if (state == 1) {
            rev.push_back(' ');
            if (!(t & E_EXT)) state = 0;
            --j;
        } else {
            rev.push_back(' ');
            if (!(t & F_EXT)) state = 0;
            --i;
        }
    }
    for (; j > 0; --j) rev.push_back(' ');
    for (; i > 0; --i) rev.push_back(' ');
    const int len = int(rev.size());
    if (len + 1 > comp_cap) return IOC_ERR_CAPACITY;
    for (int k = 0; k < len; ++k) comp[k] = rev[size_t(len - 1 - k)];
    comp[len] = 0;
    return len;
}

// setGapOpen, src/cluster.cpp:425-440
int32_t ioc_host_gap_open(double e)
{
    if (e <= 0.01) return 5;
    if (e <= 0.04) return 4;
    if (e <= 0.1) return 3;
    return 2;
}

// getAlnRatio, src/cluster.cpp:442-459
double ioc_host_aln_ratio(const char* comp, int32_t comp_len, double e, uint32_t slen, uint32_t k)
{
    if (!comp || comp_len < 0 || uint32_t(comp_len) < k || slen == 0) return 0.0;
    double aligned = 0;
    const double limit = std::floor((1.0 - e) * k);
    int nm = 0;
    for (uint32_t t = 0; t < k; ++t) nm += comp[t] == '|';
    // windows [i, i+k) for i = 0 .. len-k-1 (the reference's loop stops when j reaches end())
    for (int32_t i = 0; i + int32_t(k) < comp_len; ++i) {
        if (nm >= limit) aligned++;
        nm -= comp[i] == '|';
        nm += comp[i + int32_t(k)] == '|';
    }
    return aligned / slen;
}

}  // extern "C"
