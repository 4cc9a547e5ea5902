// ioc_align_gpu.hip — batched semi-global affine alignment on the GPU for the sahlin / furious fallback
// (getBestClusterAln, src/cluster.cpp:461-515: ParasailAlign :408-423 + getAlnRatio :442-459).
//
// The reference needs the alignment only for ONE number: how many k-windows of the comparison string
// hold at least floor((1-e)k) matches (getAlnRatio).  That number is a function of the optimal path, so
// the traceback matrix is not needed: every DP state (H, E, F of a cell) carries the statistics of ITS
// best path — the last k comparison characters as a bit window and the count of qualifying windows so
// far — and the move that wins the max also hands over its statistics.  The recurrence, the strict-`>`
// tie-breaks and the end-cell choice are those of the host aligner (ioc_align.cpp), whose traceback
// would walk exactly the moves recorded here, so (score, window count) are bit-identical to it.
// No traceback storage: 280 M cells of a 16.7 kb x 16.7 kb pair stay in registers.
//
// Mapping: one workgroup per pair, NT = 64 * waves threads.  Thread g owns ALN_C consecutive columns of
// a strip of NT * ALN_C columns and walks down the rows skewed by g (systolic wavefront): at step s it
// computes row s - g.  Its right-edge (H, E) record and the query base move to thread g + 1 by a wave
// shuffle (through LDS between waves, one barrier per step); the strip's right edge goes through a
// global scratch column to the next strip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "ioc_internal.h"

namespace {

constexpr int ALN_C = 8;             // columns per thread
constexpr int ALN_MAXW = 8;          // waves per workgroup
constexpr int ALN_NEG = INT32_MIN / 4;
constexpr uint32_t ALN_LEN_SHIFT = 26;  // window statistics word: [31:26] min(len, k), [25:0] qualifying windows

struct AlnPairDev {
    uint32_t q_off, n;   // query (rows)
    uint32_t r_off, m;   // reference (columns)
    int32_t gap_open, ilimit;
    uint32_t rc;         // reference is read reverse-complemented
    uint32_t pad;
};

struct AlnParams {
    int32_t match, mismatch, gap_extend;
    uint32_t k;     // window length, 1..32
    uint32_t mbit;  // 1 << (32 - k): where a new comparison bit enters the (top-aligned) window
};

struct St {  // one DP state: score + statistics of its best path
    int s;
    uint32_t b, c;
};

__device__ __forceinline__ uint32_t spaces_c(uint32_t g, uint32_t k, int il)
{
    // g end-gap columns (all ' '): len = min(g, k); a window of blanks qualifies iff 0 >= ilimit
    const uint32_t len = g < k ? g : k;
    const uint32_t cnt = (il <= 0 && g > k) ? g - k : 0u;
    return (len << ALN_LEN_SHIFT) | cnt;
}

// append one comparison character (bit = mbit for '|', 0 for ' ') to a path's statistics.
// hb = 64 - ilimit (ilimit clamped to 0..33): popc(window) + hb has bit 6 set iff popc >= ilimit, so the
// "window qualifies" test is two plain ALU ops and never touches VCC.
// GEN: paths shorter than k may exist (top-left k x k corner of the matrix only).
template <bool GEN>
__device__ __forceinline__ void append(uint32_t& b, uint32_t& c, uint32_t bit, uint32_t hb, uint32_t k)
{
    const uint32_t hit = (uint32_t(__popc(b)) + hb) >> 6;  // 0 or 1
    if (GEN) {
        const bool full = (c >> ALN_LEN_SHIFT) >= k;
        c += full ? hit : (1u << ALN_LEN_SHIFT);
    } else {
        c += hit;
    }
    b = (b << 1) | bit;
}

// value of the left neighbour lane (lane 0 keeps its own): one DPP move instead of an LDS permute
__device__ __forceinline__ uint32_t from_left(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ uint8_t comp_base(uint8_t ch)
{
    return ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
}

template <bool GEN>
__device__ __forceinline__ void row_cells(St (&Hp)[ALN_C], St (&F)[ALN_C], const uint32_t (&rpk)[ALN_C / 4], St& hl,
                                          St& el, St dg, uint32_t qc, int go, uint32_t hb, const AlnParams& P)
{
#pragma unroll
    for (int c = 0; c < ALN_C; ++c) {
        // E: gap in the query (horizontal move) from the cell on the left
        St E;
        {
            const int eo = hl.s - go, ee = el.s - P.gap_extend;
            const bool ex = ee > eo;
            E.s = ex ? ee : eo;
            E.b = ex ? el.b : hl.b;
            E.c = ex ? el.c : hl.c;
            append<GEN>(E.b, E.c, 0u, hb, P.k);
        }
        // F: gap in the reference (vertical move) from the cell above
        St Fn;
        {
            const int fo = Hp[c].s - go, fe = F[c].s - P.gap_extend;
            const bool fx = fe > fo;
            Fn.s = fx ? fe : fo;
            Fn.b = fx ? F[c].b : Hp[c].b;
            Fn.c = fx ? F[c].c : Hp[c].c;
            append<GEN>(Fn.b, Fn.c, 0u, hb, P.k);
        }
        St h;
        {
            const bool mt = qc == ((rpk[c >> 2] >> (8 * (c & 3))) & 0xFFu);
            h.s = dg.s + (mt ? P.match : P.mismatch);
            h.b = dg.b;
            h.c = dg.c;
            append<GEN>(h.b, h.c, mt ? P.mbit : 0u, hb, P.k);
        }
        if (E.s > h.s) h = E;
        if (Fn.s > h.s) h = Fn;
        dg = Hp[c];
        Hp[c] = h;
        F[c] = Fn;
        hl = h;
        el = E;
    }
}

__global__ void __launch_bounds__(64 * ALN_MAXW, 3)
k_align(const AlnPairDev* __restrict__ pairs, const uint32_t* __restrict__ order, const uint8_t* __restrict__ pool,
        AlnParams P, uint32_t* __restrict__ bnd, uint64_t bnd_stride, uint32_t* __restrict__ lrow,
        uint64_t lrow_stride, int32_t* __restrict__ out_score, uint32_t* __restrict__ out_count)
{
    __shared__ uint32_t xb[2][ALN_MAXW][8];
    __shared__ uint32_t s_look[7][64];
    __shared__ uint32_t s_lc[4];
    const uint32_t pid = order[blockIdx.x];
    const AlnPairDev pr = pairs[pid];
    const uint32_t n = pr.n, m = pr.m;
    const int go = pr.gap_open, il = pr.ilimit;
    const uint32_t hb = uint32_t(64 - (il < 0 ? 0 : il > 33 ? 33 : il));
    const uint8_t* __restrict__ q = pool + pr.q_off;
    const uint8_t* __restrict__ r = pool + pr.r_off;
    const uint32_t NT = blockDim.x;
    const uint32_t g = threadIdx.x, lane = g & 63u, wave = g >> 6, nwaves = NT >> 6;
    uint32_t* mybnd = bnd + uint64_t(blockIdx.x) * bnd_stride;    // [2][6][n]
    uint32_t* mylrow = lrow + uint64_t(blockIdx.x) * lrow_stride;  // [entries][4]
    const uint32_t strip_cols = NT * ALN_C;
    const uint32_t nstrips = (m + strip_cols - 1) / strip_cols;
    const uint32_t nsteps = n + NT - 1;
    if (g == 0) {
        s_lc[0] = uint32_t(ALN_NEG);
        s_lc[1] = s_lc[2] = s_lc[3] = 0;
    }

    for (uint32_t p = 0; p < nstrips; ++p) {
        const uint32_t jb = p * strip_cols + g * ALN_C;  // columns to the left of this thread's block
        uint32_t rpk[ALN_C / 4];  // this thread's reference bytes, 4 per register; 0 (never a base) beyond the end
#pragma unroll
        for (int c4 = 0; c4 < ALN_C / 4; ++c4) {
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t j = jb + c4 * 4 + e;
                uint32_t ch = 0;
                if (j < m) ch = pr.rc ? comp_base(r[m - 1 - j]) : r[j];
                w |= ch << (8 * e);
            }
            rpk[c4] = w;
        }
        St Hp[ALN_C], F[ALN_C];
#pragma unroll
        for (int c = 0; c < ALN_C; ++c) {
            Hp[c].s = 0;  // row 0: free leading gap, jb + c + 1 blank columns
            Hp[c].b = 0;
            Hp[c].c = spaces_c(jb + c + 1, P.k, il);
            F[c].s = ALN_NEG;
            F[c].b = 0;
            F[c].c = 0;
        }
        St dg{0, 0u, spaces_c(jb, P.k, il)};  // H(0, jb)
        const int lastc = (m - 1 >= jb && m - 1 < jb + ALN_C) ? int(m - 1 - jb) : -1;
        St bc{ALN_NEG, 0u, 0u};
        uint32_t bc_i = 0;
        const bool corner_cols = jb < P.k;
        const uint32_t* bin = mybnd + uint64_t((p + 1) & 1u) * 6u * n;  // written by strip p-1
        uint32_t* bout = mybnd + uint64_t(p & 1u) * 6u * n;
        const bool write_edge = (g == NT - 1) && (p + 1 < nstrips);

        // wave 0 looks ahead in blocks of 64 rows — the query bytes and (strips > 0) the left-edge records:
        // loaded one block early into registers, published to LDS when the block becomes current; lane 0
        // (the only consumer) picks one row per step
        uint32_t qn = 0, en[6] = {0, 0, 0, 0, 0, 0};
        St out_h{0, 0u, 0u}, out_e{ALN_NEG, 0u, 0u};
        uint32_t out_q = 0;
        for (uint32_t s = 0; s < nsteps; ++s) {
            if (wave == 0 && (s & 63u) == 0) {
                if (s == 0) {
                    const uint32_t row = lane;
                    qn = row < n ? q[row] : 0u;
                    if (p > 0) {
#pragma unroll
                        for (int w = 0; w < 6; ++w) en[w] = row < n ? bin[uint64_t(w) * n + row] : 0u;
                    }
                }
                s_look[6][lane] = qn;
#pragma unroll
                for (int w = 0; w < 6; ++w) s_look[w][lane] = en[w];
                const uint32_t row = s + 64u + lane;
                qn = row < n ? q[row] : 0u;
                if (p > 0) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) en[w] = row < n ? bin[uint64_t(w) * n + row] : 0u;
                }
            }
            // inputs of this step: what the left neighbour produced in the previous step
            St hl, el;
            uint32_t qc;
            hl.s = int(from_left(uint32_t(out_h.s)));
            hl.b = from_left(out_h.b);
            hl.c = from_left(out_h.c);
            el.s = int(from_left(uint32_t(out_e.s)));
            el.b = from_left(out_e.b);
            el.c = from_left(out_e.c);
            qc = from_left(out_q);
            if (lane == 0) {
                if (wave > 0) {
                    const uint32_t* x = xb[(s + 1) & 1u][wave - 1];
                    hl.s = int(x[0]);
                    hl.b = x[1];
                    hl.c = x[2];
                    el.s = int(x[3]);
                    el.b = x[4];
                    el.c = x[5];
                    qc = x[6];
                } else {
                    const uint32_t sl = s & 63u;
                    qc = s_look[6][sl];
                    if (p == 0) {
                        hl.s = 0;  // column 0: free leading gap of s + 1 blank columns
                        hl.b = 0;
                        hl.c = spaces_c(s + 1, P.k, il);
                        el.s = ALN_NEG;
                        el.b = 0;
                        el.c = 0;
                    } else {
                        hl.s = int(s_look[0][sl]);
                        hl.b = s_look[1][sl];
                        hl.c = s_look[2][sl];
                        el.s = int(s_look[3][sl]);
                        el.b = s_look[4][sl];
                        el.c = s_look[5][sl];
                    }
                }
            }
            const int i = int(s) - int(g);
            const bool active = i >= 0 && uint32_t(i) < n;
            if (active) {
                const St hl_in = hl;
                if (corner_cols && uint32_t(i) < P.k)
                    row_cells<true>(Hp, F, rpk, hl, el, dg, qc, go, hb, P);
                else
                    row_cells<false>(Hp, F, rpk, hl, el, dg, qc, go, hb, P);
                dg = hl_in;
                if (lastc >= 0) {  // one thread of the last strip: best of the last column, first row wins ties
                    St hm = Hp[0];
#pragma unroll
                    for (int c = 1; c < ALN_C; ++c)
                        if (c == lastc) hm = Hp[c];
                    if (hm.s > bc.s) {
                        bc = hm;
                        bc_i = uint32_t(i) + 1u;
                    }
                }
                if (write_edge) {
                    bout[0ull * n + uint32_t(i)] = uint32_t(hl.s);
                    bout[1ull * n + uint32_t(i)] = hl.b;
                    bout[2ull * n + uint32_t(i)] = hl.c;
                    bout[3ull * n + uint32_t(i)] = uint32_t(el.s);
                    bout[4ull * n + uint32_t(i)] = el.b;
                    bout[5ull * n + uint32_t(i)] = el.c;
                }
                if (uint32_t(i) == n - 1) {
                    // last row: this thread's best cell, first column wins ties
                    St br{ALN_NEG, 0u, 0u};
                    uint32_t bj = 0;
#pragma unroll
                    for (int c = 0; c < ALN_C; ++c) {
                        if (jb + c < m && Hp[c].s > br.s) {
                            br = Hp[c];
                            bj = jb + c + 1;
                        }
                    }
                    uint32_t* e = mylrow + uint64_t(p * NT + g) * 4u;
                    e[0] = uint32_t(br.s);
                    e[1] = bj;
                    e[2] = br.b;
                    e[3] = br.c;
                }
            }
            out_h = hl;
            out_e = el;
            out_q = qc;
            if (lane == 63 && wave + 1 < nwaves) {
                uint32_t* x = xb[s & 1u][wave];
                x[0] = uint32_t(hl.s);
                x[1] = hl.b;
                x[2] = hl.c;
                x[3] = uint32_t(el.s);
                x[4] = el.b;
                x[5] = el.c;
                x[6] = qc;
            }
            if (nwaves > 1) __syncthreads();
        }
        if (lastc >= 0) {
            s_lc[0] = uint32_t(bc.s);
            s_lc[1] = bc_i;
            s_lc[2] = bc.b;
            s_lc[3] = bc.c;
        }
        __syncthreads();
    }

    // end cell: best of the last column (rows ascending), replaced only by a strictly larger cell of the
    // last row (columns ascending from 0) — the host aligner's scan order (ioc_align.cpp)
    if (wave == 0) {
        St br{0, 0u, spaces_c(n, P.k, il)};  // H(n, 0)
        uint32_t bj = 0;
        const uint32_t entries = nstrips * NT;
        for (uint32_t e = lane; e < entries; e += 64) {
            const uint32_t* x = mylrow + uint64_t(e) * 4u;
            const int s = int(x[0]);
            const uint32_t j = x[1];
            if (s > br.s || (s == br.s && j < bj)) {
                br.s = s;
                bj = j;
                br.b = x[2];
                br.c = x[3];
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int s = __shfl_xor(br.s, o);
            const uint32_t j = __shfl_xor(bj, o), b = __shfl_xor(br.b, o), c = __shfl_xor(br.c, o);
            if (s > br.s || (s == br.s && j < bj)) {
                br.s = s;
                bj = j;
                br.b = b;
                br.c = c;
            }
        }
        if (lane == 0) {
            St fin{int(s_lc[0]), s_lc[2], s_lc[3]};
            uint32_t bi = s_lc[1], bjj = m;
            if (br.s > fin.s) {
                fin = br;
                bi = n;
                bjj = bj;
            }
            // trailing end gaps: (m - bj) + (n - bi) blanks
            const uint32_t gtrail = (m - bjj) + (n - bi);
            const uint32_t t = gtrail < P.k ? gtrail : P.k;
            for (uint32_t x = 0; x < t; ++x) append<true>(fin.b, fin.c, 0u, hb, P.k);
            uint32_t cnt = fin.c & ((1u << ALN_LEN_SHIFT) - 1u);
            if (il <= 0) cnt += gtrail - t;
            out_score[pid] = fin.s;
            out_count[pid] = cnt;
        }
    }
}

struct DevTmp {
    void* p = nullptr;
    ~DevTmp()
    {
        if (p) (void)hipFree(p);
    }
};

int reserve(ioc_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return IOC_OK;
    if (b.p) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, "stream synchronize failed");
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    const hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return ioc_fail(c, IOC_ERR_CAPACITY, "hipMalloc(" + std::to_string(want) + " B) failed: " + hipGetErrorString(e));
    }
    b.cap = want;
    return IOC_OK;
}

}  // namespace

#define ACHK(c, call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int ioc_align_set_pool(ioc_ctx* c, int32_t n_seqs, const char* seqs, const int64_t* offs)
{
    if (!c || n_seqs < 0 || (n_seqs > 0 && (!seqs || !offs))) return IOC_ERR_ARG;
    ACHK(c, hipSetDevice(c->device));
    c->aln_offs.assign(offs, offs + (n_seqs > 0 ? n_seqs + 1 : 0));
    if (n_seqs == 0) return IOC_OK;
    if (offs[0] != 0) return ioc_fail(c, IOC_ERR_ARG, "sequence pool offsets must start at 0");
    for (int32_t i = 0; i < n_seqs; ++i)
        if (offs[i + 1] < offs[i]) return ioc_fail(c, IOC_ERR_ARG, "sequence pool offsets must be ascending");
    const int64_t total = offs[n_seqs];
    if (total >= (int64_t(1) << 32)) return ioc_fail(c, IOC_ERR_CAPACITY, "sequence pool above 4 GiB");
    int r = reserve(c, c->a_pool, size_t(total));
    if (r != IOC_OK) return r;
    ACHK(c, hipMemcpyAsync(c->a_pool.p, seqs, size_t(total), hipMemcpyHostToDevice, c->stream));
    ACHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_align_pairs(ioc_ctx* c, int32_t n_pairs, const ioc_aln_pair* pairs, int32_t k, int32_t match, int32_t mismatch,
                    int32_t gap_extend, int32_t* out_score, int64_t* out_windows, double* out_ratio)
{
    if (!c || n_pairs < 0 || (n_pairs > 0 && !pairs)) return IOC_ERR_ARG;
    if (k < 1 || k > 32) return ioc_fail(c, IOC_ERR_CAPACITY, "GPU aligner: window length k must be in 1..32");
    ACHK(c, hipSetDevice(c->device));
    if (n_pairs == 0) return IOC_OK;
    const int32_t n_seqs = c->aln_offs.empty() ? 0 : int32_t(c->aln_offs.size() - 1);
    std::vector<AlnPairDev> dp;
    std::vector<uint32_t> back;  // device pair -> caller pair
    dp.reserve(size_t(n_pairs));
    uint32_t max_n = 1, max_m = 1;
    for (int32_t i = 0; i < n_pairs; ++i) {
        const ioc_aln_pair& a = pairs[i];
        if (a.query < 0 || a.query >= n_seqs || a.ref < 0 || a.ref >= n_seqs)
            return ioc_fail(c, IOC_ERR_ARG, "alignment pair refers to a sequence outside the pool");
        const int64_t n = c->aln_offs[size_t(a.query) + 1] - c->aln_offs[size_t(a.query)];
        const int64_t m = c->aln_offs[size_t(a.ref) + 1] - c->aln_offs[size_t(a.ref)];
        if (n + m >= (int64_t(1) << ALN_LEN_SHIFT))
            return ioc_fail(c, IOC_ERR_CAPACITY, "GPU aligner: sequences above 2^26 bases");
        const double limit = std::floor((1.0 - a.e) * double(k));  // getAlnRatio, cluster.cpp:446
        const int32_t il = limit < -1.0 ? -1 : limit > 64.0 ? 64 : int32_t(limit);
        if (n == 0 || m == 0) {
            // nothing to align: the comparison string is n + m blanks
            const int64_t len = n + m;
            const int64_t cnt = (il <= 0 && len > k) ? len - k : 0;
            if (out_score) out_score[i] = 0;
            if (out_windows) out_windows[i] = cnt;
            if (out_ratio) out_ratio[i] = n == 0 ? 0.0 : double(cnt) / double(n);
            continue;
        }
        AlnPairDev d{};
        d.q_off = uint32_t(c->aln_offs[size_t(a.query)]);
        d.n = uint32_t(n);
        d.r_off = uint32_t(c->aln_offs[size_t(a.ref)]);
        d.m = uint32_t(m);
        d.gap_open = ioc_host_gap_open(a.e);
        d.ilimit = il;
        d.rc = a.ref_revcomp ? 1u : 0u;
        dp.push_back(d);
        back.push_back(uint32_t(i));
        max_n = std::max(max_n, d.n);
        max_m = std::max(max_m, d.m);
    }
    const uint32_t np = uint32_t(dp.size());
    if (np == 0) return IOC_OK;
    // heaviest pairs first
    std::vector<uint32_t> order(np);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return uint64_t(dp[x].n) * dp[x].m > uint64_t(dp[y].n) * dp[y].m;
    });
    // waves per pair: few pairs -> wide workgroups (latency), many pairs -> narrow ones (no fill/drain waste)
    uint32_t waves = ALN_MAXW;
    if (const char* e = getenv("IOC_ALIGN_WAVES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= ALN_MAXW) waves = uint32_t(v);
    } else {
        while (waves > 1 && uint64_t(np) * waves > 4096) waves >>= 1;
        while (waves > 1 && uint64_t(waves / 2) * 64 * ALN_C >= max_m) waves >>= 1;
    }
    const uint32_t NT = waves * 64;
    const uint64_t bnd_stride = 2ull * 6ull * max_n;
    const uint64_t lrow_entries = uint64_t((max_m + NT * ALN_C - 1) / (NT * ALN_C)) * NT;
    const uint64_t lrow_stride = lrow_entries * 4ull;
    // scratch is per workgroup; run in slices when the whole batch would not fit the budget
    uint64_t budget = 8ull << 30;
    const uint64_t per_pair = (bnd_stride + lrow_stride) * 4ull;
    uint32_t slice = uint32_t(std::min<uint64_t>(np, std::max<uint64_t>(1, budget / per_pair)));
    int r;
    if ((r = reserve(c, c->a_pairs, size_t(np) * sizeof(AlnPairDev))) != IOC_OK) return r;
    if ((r = reserve(c, c->a_order, size_t(np) * 4)) != IOC_OK) return r;
    if ((r = reserve(c, c->a_out, size_t(np) * 8)) != IOC_OK) return r;
    if ((r = reserve(c, c->a_bnd, size_t(slice) * bnd_stride * 4)) != IOC_OK) return r;
    if ((r = reserve(c, c->a_lrow, size_t(slice) * lrow_stride * 4)) != IOC_OK) return r;
    hipStream_t s = c->stream;
    ACHK(c, hipMemcpyAsync(c->a_pairs.p, dp.data(), size_t(np) * sizeof(AlnPairDev), hipMemcpyHostToDevice, s));
    ACHK(c, hipMemcpyAsync(c->a_order.p, order.data(), size_t(np) * 4, hipMemcpyHostToDevice, s));
    AlnParams P{match, mismatch, gap_extend, uint32_t(k), 1u << (32 - k)};
    int32_t* d_score = static_cast<int32_t*>(c->a_out.p);
    uint32_t* d_count = reinterpret_cast<uint32_t*>(d_score + np);
    for (uint32_t first = 0; first < np; first += slice) {
        const uint32_t cnt = std::min(slice, np - first);
        hipLaunchKernelGGL(k_align, dim3(cnt), dim3(NT), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p),
                           static_cast<const uint32_t*>(c->a_order.p) + first, static_cast<const uint8_t*>(c->a_pool.p), P,
                           static_cast<uint32_t*>(c->a_bnd.p), bnd_stride, static_cast<uint32_t*>(c->a_lrow.p),
                           lrow_stride, d_score, d_count);
        ACHK(c, hipGetLastError());
    }
    std::vector<int32_t> hs(np);
    std::vector<uint32_t> hc(np);
    ACHK(c, hipMemcpyAsync(hs.data(), d_score, size_t(np) * 4, hipMemcpyDeviceToHost, s));
    ACHK(c, hipMemcpyAsync(hc.data(), d_count, size_t(np) * 4, hipMemcpyDeviceToHost, s));
    ACHK(c, hipStreamSynchronize(s));
    for (uint32_t x = 0; x < np; ++x) {
        const uint32_t i = back[x];
        if (out_score) out_score[i] = hs[x];
        if (out_windows) out_windows[i] = int64_t(hc[x]);
        if (out_ratio) out_ratio[i] = double(hc[x]) / double(dp[x].n);  // getAlnRatio: aligned / slen
    }
    return IOC_OK;
}

}  // extern "C"
