// ioc_align_gpu.hip — batched semi-global affine alignment on the GPU for the sahlin / furious fallback
// (getBestClusterAln, src/cluster.cpp:461-515: ParasailAlign :408-423 + getAlnRatio :442-459).
//
// The reference needs the alignment only for ONE number: how many k-windows of the comparison string
// hold at least floor((1-e)k) matches (getAlnRatio).  Two device formulations, both bit-identical to the
// host aligner (ioc_align.cpp: same recurrence, strict-`>` tie-breaks, end-cell choice):
//
//  * "trace" (default) — what parasail_sg_trace does, without its O(n*m) traceback matrix:
//      pass 1  k_align_fwd    score-only Gotoh DP (~10 VALU per cell), which also drops CHECKPOINTS: (H, F)
//                             of every 256th row and (H, E) of every 256th column, and finds the end cell;
//      pass 2  k_align_trace  walks back from the end cell one 256 x 256 tile at a time: the tile's
//                             direction nibbles are recomputed in LDS from its top-row / left-column
//                             checkpoints, the walk feeds the comparison bits (in reverse order) into the
//                             sliding window counter.  Only ~(n + m) / 256 * 2 tiles are ever recomputed.
//  * "carry" (IOC_ALIGN_VARIANT=carry) — single pass, every DP state carries the window statistics of its
//      best path (~39 VALU per cell); kept as an independent cross-check of the trace variant.
//
// Mapping of the forward passes: one workgroup per pair, NT = 64 * waves threads.  Thread g owns C
// consecutive columns of a strip of NT * C columns and walks down the rows skewed by g (systolic wavefront):
// at step s it computes row s - g.  Its right-edge (H, E) and the query base move to thread g + 1 by one DPP
// wave shift (through LDS between waves, one barrier per step).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <type_traits>
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
    // verdict mode (ioc_align_set_verdict_threshold): the traceback may stop once the count of good windows has reached stop_at
    // (the ratio is >= the threshold whatever follows) or can no longer reach it; 0 = walk to the end
    uint32_t stop_at;
    float e_sum;         // (host side only) the pair's summed error rate e1 + e2 (ioc_aln_pair::e): what the corridor is planned from
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

// same, lane 0 takes `first` (its own copy of it) instead: the strip's left edge enters the wave here
__device__ __forceinline__ uint32_t from_left_or(uint32_t v, uint32_t first)
{
    return uint32_t(__builtin_amdgcn_update_dpp(int(first), int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
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
k_align_carry(const AlnPairDev* __restrict__ pairs, const uint32_t* __restrict__ order, const uint8_t* __restrict__ pool,
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

// ---- "trace" variant ------------------------------------------------------------------------------
constexpr int FW_C = 16;      // columns per thread in the forward pass
#ifndef IOC_ALIGN_TILE
#define IOC_ALIGN_TILE 128
#endif
constexpr int TILE = IOC_ALIGN_TILE;  // checkpoint pitch = traceback tile edge (128 or 256)
static_assert(TILE == 128 || TILE == 256, "tile edge");
constexpr int TR_C = TILE / 64;  // columns per lane in the traceback tile (one wave per pair)

struct AlnCk {  // where a pair's checkpoints live in the arena, in int2 units
    uint64_t row_off;  // [(n - 1) / TILE][2][row_pitch(m)] ints: the Hq plane, then the F* plane, of rows TILE, 2 TILE, ...
    uint64_t col_off;  // [(m - 1) / TILE][col_pitch(n)]  (H, E) of columns TILE, 2 TILE, ...
};

// row checkpoints are padded to whole lane blocks (16 columns): a lane stores its block without per-column
// bounds checks — the store block runs in EVERY step of a wave (for the two lanes whose skewed row block ends a
// tile), so its instruction count matters as much as the cells'
__host__ __device__ __forceinline__ uint64_t row_pitch(uint32_t m) { return (uint64_t(m) + 15u) & ~uint64_t(15); }
// Packed forward pass (k_align_fwd16): a wave stores the row checkpoints of its two bands as they sit in its
// registers, 16-bit values relative to a base, and the base of each block of 16 columns next to them.  The two
// bands are on different strips, so each has planes of its own (of which only "its" half of a word counts).
// Entry e = wave * tiles_per_band + tile_in_band, in uint32 units: planes [Hq low band][F* low][Hq high][F* high]
// of `pitch` words, then pitch / 16 bases of the low band's blocks and pitch / 16 of the high band's (+ padding).
__host__ __device__ __forceinline__ uint64_t row16_stride(uint32_t m) { return 4u * row_pitch(m) + row_pitch(m) / 4u; }  // (16-byte multiples)
// int2 units the row region needs at most, whatever the number of waves (<= 4) a pair gets
__host__ __device__ __forceinline__ uint64_t row16_units(uint32_t n, uint32_t m)
{
    const uint64_t ntiles = (uint64_t(n) + TILE - 1) / TILE;
    return ((ntiles / 2u + 5u) * row16_stride(m) + 1u) / 2u;
}
// one value of a packed row checkpoint: band half hs of entry `ent`, plane pl (0 Hq, 1 F*), column j
__device__ __forceinline__ int row16_get(const uint32_t* __restrict__ ent, uint32_t m, int hs, int pl, uint32_t j)
{
    const uint32_t w = ent[uint64_t(2 * hs + pl) * row_pitch(m) + j];
    const int base = int(ent[4u * row_pitch(m) + uint32_t(hs) * (row_pitch(m) / 16u) + j / 16u]);
    return (hs ? (int(w) >> 16) : int(short(w & 0xFFFFu))) + base;
}
// column checkpoints: 4 rows (one step of a lane) go out as two 16-byte stores
__host__ __device__ __forceinline__ uint64_t col_pitch(uint32_t n) { return (uint64_t(n) + 3u) & ~uint64_t(3); }

__device__ __forceinline__ uint32_t ref_byte(const uint8_t* __restrict__ r, uint32_t m, uint32_t rc, uint32_t j)
{
    if (j >= m) return 0u;  // never equals a base
    return rc ? comp_base(r[m - 1 - j]) : r[j];
}

// The forward pass works on SLANTED scores: X*(i, j) = X(i, j) + ge * (i + j) for X in {H, E, F}, and keeps
// Hq = H* - (go - ge) instead of H*.  Extending a gap then costs nothing (the slant pays the ge), opening
// one is the "- (go - ge)" already folded into Hq, and a diagonal step adds match + 2 ge or mismatch + 2 ge:
//     E*(i, j) = max(E*(i, j-1), Hq(i, j-1))          F*(i, j) = max(F*(i-1, j), Hq(i-1, j))
//     H*(i, j) = max3(Hq(i-1, j-1) + (s + 2 ge + go - ge), E*, F*)
// 7 VALU per cell instead of 10; true values (checkpoints, end cell) are X* - ge * (i + j).
// tail launch of the forward pass: a pair split over `groups` workgroups (see k_align_fwd)
struct AlnCross {
    uint32_t groups = 1;       // workgroups per pair (1: the usual launch)
    uint32_t first_wg = 0;     // first workgroup of the launch that belongs to a split pair
    uint32_t first_pair = 0;   // ... and the pair (position in `order`) it starts with
    uint32_t flag_stride = 0;  // flags per pair
    uint32_t* flags = nullptr; // [pair][band][strip]: the band's checkpoints of the strip are out
    int2* best = nullptr;      // [pair][band]: best cell of the band's part of the last column
    uint32_t* err = nullptr;   // a wait ran out
};

struct FwdConst {
    int gd;      // go - ge
    int cm, cx;  // match / mismatch + 2 ge + gd
    int ge;
};

// An opaque copy of a per-lane value: address / predicate / slant arithmetic derived from it inside a
// rarely taken block stays inside that block instead of being hoisted out of the step loop into dozens of
// long-lived registers.
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// PROF: the diagonal increment (cm for identical bases, cx otherwise) comes as one byte per column out of
// the lane's QUERY PROFILE in LDS — per strip, for each of A C G T (and "anything else"), the 16 increments
// of the lane's 16 columns packed four to a word; `xw` holds the row of this query base.  One add with a
// byte-select operand instead of compare + select + add: 5 VALU per cell.
template <int C, bool PROF>
__device__ __forceinline__ void fwd_cells(int (&Hq)[C], int (&F)[C], const uint32_t (&rpk)[C / 4], int& hql, int& el,
                                          int dgq, uint32_t qc, const FwdConst& K)
{
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int E = max(el, hql);
        const int Fn = max(F[c], Hq[c]);
        int inc;
        if (PROF) {
            inc = int((rpk[c >> 2] >> (8 * (c & 3))) & 0xFFu);  // rpk = the profile row here
        } else {
            const bool mt = qc == ((rpk[c >> 2] >> (8 * (c & 3))) & 0xFFu);
            inc = mt ? K.cm : K.cx;
        }
        const int hq = max(max(dgq + inc, E), Fn) - K.gd;
        dgq = Hq[c];
        Hq[c] = hq;
        F[c] = Fn;
        hql = hq;
        el = E;
    }
}

// Pass 1: scores only.  A strip is 64 * FW_C = 1024 columns (a multiple of TILE, so its right edge IS a
// column checkpoint and the next strip reads its left edge from there); one WAVE walks a strip top to
// bottom, FW_R rows per step, its lanes skewed by one step: the right edge (Hq, E*) and the query bases
// of a lane reach the next lane by one DPP wave shift — no LDS, no barrier inside a strip.
// The waves of a workgroup split the ROWS into bands of whole tiles and run the (band, strip) grid as a
// pipeline: wave b does strip p in round p + b, taking its top edge from the row checkpoint band b - 1
// wrote one round earlier.  One barrier per round (~4000 steps) instead of one per step: waves of one pair
// share SIMDs with other pairs' waves, and per-step lockstep cost 30 % of the throughput.
constexpr int FW_R = 4;
static_assert(FW_C == 16, "the last-column select tree assumes 16 (or, narrow last strip, 8) columns per lane");
static_assert(FW_R == 4, "the look-ahead hands 4 rows per step to lane 0 (one 16-byte LDS read per field)");

#ifndef IOC_FWD_WAVES_PER_EU
#define IOC_FWD_WAVES_PER_EU 3
#endif
#ifndef IOC_FWD_NARROW_LAST
#define IOC_FWD_NARROW_LAST 1
#endif
#ifndef IOC_TR_STEP_UNROLL
#define IOC_TR_STEP_UNROLL 2
#endif
#ifndef IOC_FWD_STEP_UNROLL
#define IOC_FWD_STEP_UNROLL 4
#endif
template <bool PROF>
__global__ void __launch_bounds__(64 * ALN_MAXW) __attribute__((amdgpu_waves_per_eu(IOC_FWD_WAVES_PER_EU, 8)))
k_align_fwd(const AlnPairDev* __restrict__ pairs, const uint32_t* __restrict__ order, uint32_t count, uint32_t wpp_main,
            uint32_t n_main, uint32_t wpp_tail, const uint8_t* __restrict__ pool, AlnParams P, int2* ck, const AlnCk* __restrict__ cko, int2* lrow,
            uint64_t lrow_stride, int4* __restrict__ ends, AlnCross X)
{
    // A workgroup is 4 (or 8) waves = one per SIMD of its CU, however the dispatcher places workgroups; it
    // carries (waves / wpp) pairs, each split over wpp waves ("bands").  (Workgroups of 2 waves were seen
    // sharing SIMDs while others idled.)
    __shared__ __attribute__((aligned(16))) uint32_t s_look_all[ALN_MAXW][3][128];
    __shared__ int s_best[ALN_MAXW][2];
    __shared__ uint32_t s_rounds;
    // PROF: query profile of the current strip, [wave][base code 0..4][lane][4 words]
    __shared__ __attribute__((aligned(16))) uint32_t s_prof_all[PROF ? ALN_MAXW : 1][5][64][4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));  // uniform: keep the band in SGPRs
    // The first n_main workgroups (one resident generation of the chip) split their pairs over wpp_main waves;
    // the ones after them — dispatched as those retire — over wpp_tail >= wpp_main waves, so that the tail
    // generation, which runs on a half-empty chip, is over sooner.
    const bool tail = blockIdx.x >= n_main;
    const uint32_t wpp = tail ? wpp_tail : wpp_main, wg_waves = blockDim.x >> 6;
    // X.groups > 1 (the tail launch): a pair is split over X.groups WORKGROUPS of 4 waves, band = 4 * group + wave;
    // the band below another workgroup's last band takes its top edges when that band's flag says they are there
    // (X.first_wg: the workgroups before it are ordinary ones of the same launch — the split pairs then start on
    // whatever CU frees up first; a workgroup only ever waits for one with a smaller index, dispatched before it)
    const bool cross = X.groups > 1 && blockIdx.x >= X.first_wg;
    const uint32_t xb = cross ? blockIdx.x - X.first_wg : 0u;
    const uint32_t grp = cross ? xb % X.groups : 0u;
    const uint32_t slot = cross ? 0u : wv / wpp, wave = cross ? grp * wg_waves + wv : wv % wpp, nwaves = cross ? X.groups * wg_waves : wpp;
    const uint32_t pslot = cross ? X.first_pair + xb / X.groups
                         : tail  ? n_main * (wg_waves / wpp_main) + (blockIdx.x - n_main) * (wg_waves / wpp_tail) + slot
                                 : blockIdx.x * (wg_waves / wpp_main) + slot;  // pair of this wave, in `order`
    const bool live = pslot < count;
    if (threadIdx.x == 0) s_rounds = 0;
    __syncthreads();
    const uint32_t pid = order[live ? pslot : 0];
    const AlnPairDev pr = pairs[pid];
    const uint32_t n = pr.n, m = pr.m;
    FwdConst K;
    K.ge = P.gap_extend;
    K.gd = pr.gap_open - P.gap_extend;
    K.cm = P.match + 2 * P.gap_extend + K.gd;
    K.cx = P.mismatch + 2 * P.gap_extend + K.gd;
    const uint8_t* __restrict__ q = pool + pr.q_off;
    const uint8_t* __restrict__ r = pool + pr.r_off;
    int2* rowck = ck + cko[pid].row_off;
    int2* colck = ck + cko[pid].col_off;
    int2* mylrow = lrow + uint64_t(live ? pslot : 0) * lrow_stride;
    uint32_t(*s_look)[128] = s_look_all[wv];
    uint32_t(*s_prof)[64][4] = s_prof_all[PROF ? wv : 0];
    // what travels with a row: the query byte, or (PROF) the word offset of its profile row
    auto qcode = [](uint32_t ch) -> uint32_t {
        if (!PROF) return ch;
        const uint32_t code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
        return code * 256u;
    };
    constexpr uint32_t strip_cols = 64u * FW_C;
    const uint32_t nstrips = (m + strip_cols - 1) / strip_cols;
    const bool narrow_last = IOC_FWD_NARROW_LAST && m - (nstrips - 1u) * strip_cols <= strip_cols / 2u;  // (uniform) see `strip`
    // this wave's band of rows [r_lo, r_hi), whole tiles
    const uint32_t ntiles = (n + TILE - 1) / TILE, tpb = (ntiles + nwaves - 1) / nwaves;
    const uint32_t r_lo = min(n, wave * tpb * TILE), r_hi = min(n, (wave + 1u) * tpb * TILE);
    const uint32_t nblocks = (r_hi - r_lo + FW_R - 1) / FW_R;  // row blocks of the band
    const uint32_t nsteps = nblocks + 63u;
    const bool last_band = r_hi == n && r_lo < n;
    int bc = ALN_NEG;  // best of the last column inside this band (true score), first row wins ties
    uint32_t bc_i = 0;

    // the pairs of a workgroup may need different numbers of rounds: everybody stays for the barriers
    uint32_t* xflag = cross ? X.flags + uint64_t(live ? pslot - X.first_pair : 0) * X.flag_stride : nullptr;  // [band][strip]
    bool xbad = false;
    if (live && lane == 0) atomicMax(&s_rounds, nstrips + nwaves - 1u);
    __syncthreads();
    const uint32_t rounds = s_rounds;
    for (uint32_t round = 0; round < rounds; ++round) {
        const int ps = int(round) - int(wave);
        if (live && ps >= 0 && uint32_t(ps) < nstrips && nblocks > 0) {
            const uint32_t p = uint32_t(ps);
            if (cross && wv == 0 && wave > 0) {
                // the band above lives in another workgroup: wait for its row checkpoint of this strip (bounded:
                // the launch keeps all its workgroups resident, but a wait without an end could take the GPU down)
                uint32_t seen = 0;
                if (lane == 0 && !xbad) {
                    for (uint32_t it = 0; it < (1u << 22); ++it) {
                        seen = __hip_atomic_load(&xflag[uint64_t(wave - 1u) * nstrips + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen) break;
                        if ((it & 1023u) == 1023u && __hip_atomic_load(X.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // somebody gave up
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (!seen) atomicOr(X.err, 1u);
                }
                seen = uint32_t(__builtin_amdgcn_readfirstlane(int(seen)));
                if (!seen) xbad = true;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            // The strip with C columns per lane: 16, or — the last strip of a pair when no more than half a strip of
            // columns is left — 8: a step then costs 45 + 4 x 8 x 5 instructions instead of 45 + 4 x 16 x 5 for the same
            // number of steps (m = 16.6 kb: 17 strips, the last one 283 columns wide on average — 18 of 64 lanes busy).
            auto strip = [&](auto cc) __attribute__((always_inline)) {
            constexpr int C = decltype(cc)::value;
            const uint32_t jb = p * strip_cols + lane * uint32_t(C);  // columns to the left of this lane's block
            uint32_t rpk[C / 4];
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
                uint32_t w = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) w |= ref_byte(r, m, pr.rc, jb + c4 * 4 + e) << (8 * e);
                rpk[c4] = w;
            }
            if (PROF) {
                const uint32_t bases = 'A' | ('C' << 8) | ('G' << 16) | ('T' << 24);
#pragma unroll
                for (int code = 0; code < 5; ++code) {
                    const uint32_t b = code < 4 ? (bases >> (8 * code)) & 0xFFu : 0x100u;  // 0x100: equals no byte
#pragma unroll
                    for (int c4 = 0; c4 < C / 4; ++c4) {
                        uint32_t w = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            w |= uint32_t(((rpk[c4] >> (8 * e)) & 0xFFu) == b ? K.cm : K.cx) << (8 * e);
                        s_prof[code][lane][c4] = w;
                    }
                }
            }
            // Hq and F* of the row above the band (slanted, see fwd_cells): row 0 of the matrix (free leading
            // gap, H = 0) or the row checkpoint the band above wrote in the previous round
            int Hp[C], F[C];
            int dg;
            if (r_lo == 0) {
                const int base = K.ge * int(jb) - K.gd;  // per-column parts (ge * c) are scalar
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    Hp[c] = base + K.ge * (c + 1);
                    F[c] = ALN_NEG;
                }
                dg = base;
            } else {
                const int* roh = reinterpret_cast<const int*>(rowck + uint64_t(r_lo / TILE - 1) * row_pitch(m));
                const int* rof = roh + row_pitch(m);
#pragma unroll
                for (int c = 0; c < C; ++c) {  // row checkpoints hold the slanted (Hq, F*)
                    const int2 v = (jb + c < m) ? int2{roh[jb + c], rof[jb + c]} : int2{0, 0};
                    Hp[c] = v.x;
                    F[c] = v.y;
                }
                dg = (jb > 0 && jb <= m) ? roh[jb - 1] : K.ge * int(r_lo) - K.gd;  // column 0 holds H = 0
            }
            const bool has_cols = jb < m;
            const int lastc = (m - 1 >= jb && m - 1 < jb + uint32_t(C)) ? int(m - 1 - jb) : -1;
            const bool strip_has_lastc = uint64_t(p + 1u) * strip_cols >= m;  // (uniform) the last strip
            // this lane's right edge is a column checkpoint (lane 63: the next strip's input)
            const uint32_t jr = jb + uint32_t(C);
            const bool wr_col = (jr % TILE) == 0 && jr < m;
            int2* colout = wr_col ? colck + uint64_t(jr / TILE - 1) * col_pitch(n) : colck;
            const int2* colin = p ? colck + uint64_t(p * strip_cols / TILE - 1) * col_pitch(n) : colck;
            // look-ahead in blocks of 64 rows — the query bytes and the strip's left edge (Hq, E*): column 0
            // of the matrix (H = 0, no gap to extend) or the column checkpoint, slanted
            auto left_edge = [&](uint32_t row) {
                int2 v{K.ge * int(row + 1) - K.gd, ALN_NEG};
                if (p > 0 && row < r_hi) v = colin[row];  // checkpoints hold the slanted (Hq, E*) as passed between lanes
                return v;
            };
            uint32_t qn = 0;
            int2 en{0, 0};
            int hl[FW_R], el[FW_R];  // inputs of the step from the left; after the step: this lane's right edge
            uint32_t qc[FW_R];
#pragma unroll
            for (int rr = 0; rr < FW_R; ++rr) {
                hl[rr] = 0;
                el[rr] = ALN_NEG;
                qc[rr] = 0;
            }
            // Look-ahead of the left edge + query, 64 rows at a time, one block AHEAD of its use in a 128-row
            // ring in LDS; every lane reads the 4 rows of the NEXT step from there (one address for the whole
            // wave, a step before they are needed) and lane 0 picks them up as the `old` operand of the DPP shift.
            auto publish = [&](uint32_t blk) {
                s_look[0][(blk & 1u) * 64u + lane] = uint32_t(en.x);
                s_look[1][(blk & 1u) * 64u + lane] = uint32_t(en.y);
                s_look[2][(blk & 1u) * 64u + lane] = qn;
            };
            auto fetch = [&](uint32_t blk) {
                const uint32_t row = r_lo + blk * 64u + lane;
                qn = qcode(row < r_hi ? q[row] : 0u);
                en = left_edge(row);
            };
            fetch(0);
            publish(0);
            fetch(1);
            uint4 nxh, nxe, nxq;  // inputs of lane 0 for the coming step
            {
                nxh = *reinterpret_cast<const uint4*>(&s_look[0][0]);
                nxe = *reinterpret_cast<const uint4*>(&s_look[1][0]);
                nxq = *reinterpret_cast<const uint4*>(&s_look[2][0]);
            }
            // one step of the wave; the loop below runs two per iteration (the compiler cannot unroll a loop with wave-level
            // operations by itself): half of the register copies that carry the inputs from step to step go away
            auto step = [&](const uint32_t s) __attribute__((always_inline)) {
                if ((s & (64 / FW_R - 1)) == 0) {  // steps 16 b .. : publish block b + 1, fetch block b + 2
                    const uint32_t blk = s / (64 / FW_R) + 1u;
                    publish(blk);
                    fetch(blk + 1u);
                }
                {
                    const uint32_t h4[FW_R] = {nxh.x, nxh.y, nxh.z, nxh.w}, e4[FW_R] = {nxe.x, nxe.y, nxe.z, nxe.w},
                                   q4[FW_R] = {nxq.x, nxq.y, nxq.z, nxq.w};
#pragma unroll
                    for (int rr = 0; rr < FW_R; ++rr) {
                        hl[rr] = int(from_left_or(uint32_t(hl[rr]), h4[rr]));
                        el[rr] = int(from_left_or(uint32_t(el[rr]), e4[rr]));
                        qc[rr] = from_left_or(qc[rr], q4[rr]);
                    }
                    const uint32_t nx = ((s + 1u) * FW_R) & 127u;  // rows of the next step
                    nxh = *reinterpret_cast<const uint4*>(&s_look[0][nx]);
                    nxe = *reinterpret_cast<const uint4*>(&s_look[1][nx]);
                    nxq = *reinterpret_cast<const uint4*>(&s_look[2][nx]);
                }
                const int bi = int(s) - int(lane);  // row block of this lane in this step
                if (bi >= 0 && uint32_t(bi) < nblocks && has_cols) {
                    const uint32_t i0 = r_lo + uint32_t(bi) * FW_R;
                    // rows past the end (last block of the last band) run on a query byte of 0 and are never looked at
                    // wave-uniform: the special path is right for every lane (its extras are guarded per lane), and a wave whose
                    // lanes disagree would run BOTH copies of the cells — in the last strip, where one lane owns the last
                    // column, for every step of the strip
                    // (in scalar terms: a lane of the strip owns the last column, or some lane — s - lane for a lane 0..63 — is on
                    // the last row block of the last band)
                    const bool special = strip_has_lastc || (last_band && s + 1u >= nblocks && s + 1u - nblocks < 64u);
                    if (!special) {
#pragma unroll
                        for (int rr = 0; rr < FW_R; ++rr) {
                            const int hl_in = hl[rr];
                            if (PROF) {
                                const uint4 x = *reinterpret_cast<const uint4*>(&s_prof[0][0][0] + qc[rr] + lane * 4u);
                                const uint32_t x4[4] = {x.x, x.y, x.z, x.w};
                                uint32_t xw[C / 4];
#pragma unroll
                                for (int c4 = 0; c4 < C / 4; ++c4) xw[c4] = x4[c4];
                                fwd_cells<C, true>(Hp, F, xw, hl[rr], el[rr], dg, qc[rr], K);
                            } else {
                                fwd_cells<C, false>(Hp, F, rpk, hl[rr], el[rr], dg, qc[rr], K);
                            }
                            dg = hl_in;
                        }
                    } else {
#pragma unroll
                        for (int rr = 0; rr < FW_R; ++rr) {
                            const int hl_in = hl[rr];
                            if (PROF) {
                                const uint4 x = *reinterpret_cast<const uint4*>(&s_prof[0][0][0] + qc[rr] + lane * 4u);
                                const uint32_t x4[4] = {x.x, x.y, x.z, x.w};
                                uint32_t xw[C / 4];
#pragma unroll
                                for (int c4 = 0; c4 < C / 4; ++c4) xw[c4] = x4[c4];
                                fwd_cells<C, true>(Hp, F, xw, hl[rr], el[rr], dg, qc[rr], K);
                            } else {
                                fwd_cells<C, false>(Hp, F, rpk, hl[rr], el[rr], dg, qc[rr], K);
                            }
                            dg = hl_in;
                            const uint32_t i = i0 + rr;  // 0-based row
                            if (i < r_hi) {
                                if (lastc >= 0) {
                                    // Hp[lastc] by a select tree over the bits of lastc (4 conditions, not 15)
                                    const uint32_t lc = opaque(uint32_t(lastc));
                                    // (written out: an index computed in a loop makes the compiler move Hp to LDS)
                                    const bool b0 = lc & 1u, b1 = lc & 2u, b2 = lc & 4u, b3 = lc & 8u;
                                    const int s0 = b0 ? Hp[1] : Hp[0], s1 = b0 ? Hp[3] : Hp[2], s2 = b0 ? Hp[5] : Hp[4],
                                              s3 = b0 ? Hp[7] : Hp[6];
                                    const int u0 = b1 ? s1 : s0, u1 = b1 ? s3 : s2;
                                    int hm = b2 ? u1 : u0;
                                    if constexpr (C == 16) {
                                        const int s4 = b0 ? Hp[9] : Hp[8], s5 = b0 ? Hp[11] : Hp[10], s6 = b0 ? Hp[13] : Hp[12],
                                                  s7 = b0 ? Hp[15] : Hp[14];
                                        const int u2 = b1 ? s5 : s4, u3 = b1 ? s7 : s6;
                                        const int v1 = b2 ? u3 : u2;
                                        hm = b3 ? v1 : hm;
                                    }
                                    hm += K.gd - K.ge * int(opaque(i) + 1 + m);  // true H(i + 1, m)
                                    if (hm > bc) {
                                        bc = hm;
                                        bc_i = i + 1u;
                                    }
                                }
                                if (i + 1u == n) {  // last row: this lane's best cell, first column wins ties
                                    int br = ALN_NEG;
                                    uint32_t bj = 0;
                                    const uint32_t jbo = opaque(jb);
                                    const int base = K.gd - K.ge * int(n + jbo);
#pragma unroll
                                    for (int c = 0; c < C; ++c) {
                                        const int ht = Hp[c] + base - K.ge * (c + 1);  // true H(n, jb + c + 1)
                                        if (jbo + c < m && ht > br) {
                                            br = ht;
                                            bj = jbo + c + 1;
                                        }
                                    }
                                    mylrow[p * 64u + lane] = int2{br, int(bj)};
                                }
                            }
                        }
                    }
                    if (wr_col) {
                        if (!special) {  // i0 is a multiple of 4: two 16-byte stores
                            int4* co = reinterpret_cast<int4*>(colout + i0);
                            co[0] = int4{hl[0], el[0], hl[1], el[1]};
                            co[1] = int4{hl[2], el[2], hl[3], el[3]};
                        } else {
#pragma unroll
                            for (int rr = 0; rr < FW_R; ++rr)
                                if (i0 + rr < r_hi) colout[i0 + rr] = int2{hl[rr], el[rr]};
                        }
                    }
                    const uint32_t i1 = i0 + FW_R;  // DP index of the block's last row (TILE is a multiple of FW_R)
                    if ((i1 % TILE) == 0 && i1 < n) {
                        const uint32_t jbo = opaque(jb);
                        // two planes (Hq, F*): each lane's 16 values go out as they sit in its registers.  The offset from the
                        // pair's (uniform) row region is a 32-bit number of ints (the launcher refuses pairs whose region is
                        // larger): scalar base + one vector offset instead of 64-bit vector address arithmetic in a block
                        // that two lanes of every wave take in every step
                        const uint32_t pitch32 = uint32_t(row_pitch(m));
                        int* rb = reinterpret_cast<int*>(rowck) + ((i1 / TILE - 1u) * (2u * pitch32) + jbo);
                        int4* roh = reinterpret_cast<int4*>(rb);
                        int4* rof = reinterpret_cast<int4*>(rb + pitch32);
#pragma unroll
                        for (int c = 0; c < C; c += 4) {
                            roh[c / 4] = int4{Hp[c], Hp[c + 1], Hp[c + 2], Hp[c + 3]};
                            rof[c / 4] = int4{F[c], F[c + 1], F[c + 2], F[c + 3]};
                        }
                    }
                }
            };
            {
                uint32_t s = 0;
#if IOC_FWD_STEP_UNROLL == 8
                for (; s + 7u < nsteps; s += 8) {
                    step(s);
                    step(s + 1u);
                    step(s + 2u);
                    step(s + 3u);
                    step(s + 4u);
                    step(s + 5u);
                    step(s + 6u);
                    step(s + 7u);
                }
#endif
#if IOC_FWD_STEP_UNROLL >= 4
                for (; s + 3u < nsteps; s += 4) {
                    step(s);
                    step(s + 1u);
                    step(s + 2u);
                    step(s + 3u);
                }
#endif
#if IOC_FWD_STEP_UNROLL >= 2
                for (; s + 1u < nsteps; s += 2) {
                    step(s);
                    step(s + 1u);
                }
#endif
                for (; s < nsteps; ++s) step(s);
            }
            // the lane that owns the last column hands the band's best to lane 0 (a strip of the last round)
            if (lastc >= 0) {
                s_best[wv][0] = bc;
                s_best[wv][1] = int(bc_i);
                if (cross) X.best[uint64_t(pslot - X.first_pair) * nwaves + wave] = int2{bc, int(bc_i)};
            }
            };
            if (narrow_last && p + 1u == nstrips)
                strip(std::integral_constant<int, 8>{});
            else
                strip(std::integral_constant<int, FW_C>{});
            if (cross) {  // this band's checkpoints of the strip are out: tell the band below (and the final reduction)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_store(&xflag[uint64_t(wave) * nstrips + p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (rounds > 1 || nwaves > 1) __syncthreads();  // orders this round's checkpoints before the next round reads them
    }
    __syncthreads();

    // end cell: best of the last column (rows ascending), replaced only by a strictly larger cell of the
    // last row (columns ascending from 0) — the host aligner's scan order (ioc_align.cpp)
    if (xbad && lane == 0) atomicOr(X.err, 1u);
    const bool reducer = cross ? (grp + 1u == X.groups && wv == 0) : wave == 0;
    if (live && reducer) {
        if (cross) {  // every band that has rows has finished its last strip?
            uint32_t okf = 1;
            if (lane == 0) {
                for (uint32_t b2 = 0; b2 < nwaves && okf; ++b2) {
                    if (min(n, b2 * tpb * TILE) >= n) break;
                    uint32_t seen = 0;
                    for (uint32_t it = 0; it < (1u << 22); ++it) {
                        seen = __hip_atomic_load(&xflag[uint64_t(b2) * nstrips + (nstrips - 1u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (!seen) okf = 0;
                }
                if (!okf) atomicOr(X.err, 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        int br = 0;  // H(n, 0)
        uint32_t bj = 0;
        for (uint32_t e = lane; e < nstrips * 64u; e += 64) {
            {   // entry e = (strip, lane): lanes right of the matrix wrote nothing
                const uint32_t ep = e >> 6, el = e & 63u;
                const uint32_t cpl = (narrow_last && ep + 1u == nstrips) ? 8u : uint32_t(FW_C);
                if (uint64_t(ep) * strip_cols + el * cpl >= m) continue;
            }
            const int2 x = mylrow[e];
            if (x.x > br || (x.x == br && uint32_t(x.y) < bj)) {
                br = x.x;
                bj = uint32_t(x.y);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int s = __shfl_xor(br, o);
            const uint32_t j = __shfl_xor(bj, o);
            if (s > br || (s == br && j < bj)) {
                br = s;
                bj = j;
            }
        }
        if (lane == 0) {
            int fin = ALN_NEG;
            uint32_t bi = 0, bjj = m;
            for (uint32_t b2 = 0; b2 < nwaves; ++b2) {  // bands top to bottom: the first row wins ties
                if (min(n, b2 * tpb * TILE) >= n) break;
                const int2 bb = cross ? X.best[uint64_t(pslot - X.first_pair) * nwaves + b2] : int2{s_best[slot * wpp + b2][0], s_best[slot * wpp + b2][1]};
                if (bb.x > fin) {
                    fin = bb.x;
                    bi = uint32_t(bb.y);
                }
            }
            if (br > fin) {
                fin = br;
                bi = n;
                bjj = bj;
            }
            ends[pid] = int4{fin, int(bi), int(bjj), 0};
        }
    }
}

// =====================================================================================================
// Pass 1, packed: the same pipeline with TWO row bands per wave, one in each 16-bit half of every register
// (v_pk_add_i16 / v_pk_max_i16 run at the rate of their 32-bit cousins: two cells per lane and instruction).
// Wave w owns bands 2w (low halves) and 2w + 1 (high halves); in round t the low half is on strip t - 2w and
// the high half one strip behind — on the strip whose bottom row the low half wrote as a row checkpoint one
// round earlier.  Everything that leaves the wave (checkpoints, end cells) is a full 32-bit value, so pass 2
// and the results are those of the 32-bit kernel bit for bit.
//
// 16 bits hold a score RELATIVE to a base that is uniform in the wave (one per half) and moves with the sweep:
// every 16 steps the base is advanced by the value of a live lane and all registers are rebased.  What is live
// in a wave at one time — 64 lanes x 16 columns, rows a few hundred apart — differs by a few thousand at most:
// two cells of that window are joined through a common ancestor by gaps / diagonals no longer than the window,
// and the slant adds ge per row and column.  A guard checks |relative H| < 28000 at every rebase and flags the
// pair (ends.w) for the 32-bit kernel otherwise.  "No score" (E at column 0, F at row 0) is -32768: it only
// ever enters a max, and the rebase subtracts with saturation.
// =====================================================================================================
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ v2s pmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2s pmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s psubs(v2s a, v2s b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ v2s pack2(int lo, int hi) { return v2s{short(lo), short(hi)}; }
template <int H>
__device__ __forceinline__ int half_of(v2s x)
{
    return H ? int(x.y) : int(x.x);
}
__device__ __forceinline__ int half_of_rt(int word, int h) { return h ? (word >> 16) : int(short(word & 0xFFFF)); }
__device__ __forceinline__ v2s dpp_left_or(v2s v, v2s first) { return as_v2s(from_left_or(as_u32(v), as_u32(first))); }

constexpr int P16_GUARD = 28000;

// one row of 16 columns, both halves: 7 packed VALU per column (the 32-bit kernel's 5, without a max3, plus
// the byte pair of the two profiles).  xl / xh: this row's profile words of the low / high half's strip.
// Dependent packed instructions cost a wait state each, so everything that does not hang on the left
// neighbour — T = max(diagonal + increment, F*) — is computed for the 16 columns first; what is left in the
// chain along the row is E* = max(E*, Hq), H* = max(T, E*), Hq = H* - gd.
__device__ __forceinline__ void fwd16_cells(v2s (&Hq)[FW_C], v2s (&F)[FW_C], const uint32_t (&xl)[FW_C / 4],
                                            const uint32_t (&xh)[FW_C / 4], v2s& hql, v2s& el, v2s dgq, v2s gd2)
{
    v2s T[FW_C];
#pragma unroll
    for (int c = 0; c < FW_C; ++c) {
        const v2s Fn = pmax(F[c], Hq[c]);
        // bytes (c & 3) of xl and xh, zero-extended into the two halves
        const uint32_t sel = 0x0c000c00u | uint32_t(4 + (c & 3)) << 16 | uint32_t(c & 3);
        const v2s inc = as_v2s(__builtin_amdgcn_perm(xh[c >> 2], xl[c >> 2], sel));
        T[c] = pmax((c ? Hq[c - 1] : dgq) + inc, Fn);
        F[c] = Fn;
    }
    // (Hq[c - 1] above is still the row before: the chain below overwrites Hq only now)
    v2s e = el, h = hql;
#pragma unroll
    for (int c = 0; c < FW_C; ++c) {
        e = pmax(e, h);
        h = pmax(T[c], e) - gd2;
        Hq[c] = h;
    }
    hql = h;
    el = e;
}

constexpr int P16_MAXW = 4;  // waves per workgroup (8 bands): the two profile buffers of a wave are 10 KB
__global__ void __launch_bounds__(64 * P16_MAXW) __attribute__((amdgpu_waves_per_eu(IOC_FWD_WAVES_PER_EU, 8)))
k_align_fwd16(const AlnPairDev* __restrict__ pairs, const uint32_t* __restrict__ order, uint32_t count, uint32_t wpp_main,
              uint32_t n_main, uint32_t wpp_tail, const uint8_t* __restrict__ pool, AlnParams P, int2* ck, const AlnCk* __restrict__ cko,
              int2* lrow, uint64_t lrow_stride, int4* __restrict__ ends)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_look_all[P16_MAXW][3][128];
    __shared__ int s_best[2 * P16_MAXW][2];
    __shared__ uint32_t s_rounds;
    __shared__ uint32_t s_ovf[P16_MAXW];
    // query profiles of two strips (strip & 1), [wave][buffer][base code 0..4][lane][4 words]
    __shared__ __attribute__((aligned(16))) uint32_t s_prof_all[P16_MAXW][2][5][64][4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    const bool tail = blockIdx.x >= n_main;
    const uint32_t wpp = tail ? wpp_tail : wpp_main, wg_waves = blockDim.x >> 6;
    const uint32_t slot = wv / wpp, wave = wv % wpp, nwaves = wpp, nbands = 2u * wpp;
    const uint32_t pslot = tail ? n_main * (wg_waves / wpp_main) + (blockIdx.x - n_main) * (wg_waves / wpp_tail) + slot
                                : blockIdx.x * (wg_waves / wpp_main) + slot;
    const bool live = pslot < count;
    if (threadIdx.x == 0) s_rounds = 0;
    if (lane == 0) s_ovf[wv] = 0;
    __syncthreads();
    const uint32_t pid = order[live ? pslot : 0];
    const AlnPairDev pr = pairs[pid];
    const uint32_t n = pr.n, m = pr.m;
    const int ge = P.gap_extend, gd = pr.gap_open - P.gap_extend;
    const int cm = P.match + 2 * ge + gd, cx = P.mismatch + 2 * ge + gd;
    const v2s gd2 = pack2(gd, gd);
    const uint8_t* __restrict__ q = pool + pr.q_off;
    const uint8_t* __restrict__ r = pool + pr.r_off;
    int2* rowck = ck + cko[pid].row_off;
    int2* colck = ck + cko[pid].col_off;
    int2* mylrow = lrow + uint64_t(live ? pslot : 0) * lrow_stride;
    uint32_t(*s_look)[128] = s_look_all[wv];
    // byte offset of a query base's profile row
    auto qcode = [](uint32_t ch) -> uint32_t {
        const uint32_t code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
        return code * 1024u;
    };
    constexpr uint32_t strip_cols = 64u * FW_C;
    const uint32_t nstrips = (m + strip_cols - 1) / strip_cols;
    const uint32_t ntiles = (n + TILE - 1) / TILE, tpb = (ntiles + nbands - 1) / nbands;
    uint32_t r_lo[2], r_hi[2], nblk[2];
    bool last_band[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t band = 2u * wave + uint32_t(h);
        r_lo[h] = min(n, band * tpb * TILE);
        r_hi[h] = min(n, (band + 1u) * tpb * TILE);
        nblk[h] = (r_hi[h] - r_lo[h] + FW_R - 1) / FW_R;
        last_band[h] = r_hi[h] == n && r_lo[h] < n;
    }
    const uint32_t nsteps = nblk[0] + 63u;  // (bands are equal but for the trailing ones: nblk[1] <= nblk[0])
    int bc[2] = {ALN_NEG, ALN_NEG};  // best of the last column inside each band (true score), first row wins ties
    uint32_t bc_i[2] = {0, 0};
    uint32_t ovf = 0;

    if (live && lane == 0) atomicMax(&s_rounds, nstrips + nbands - 1u);
    __syncthreads();
    const uint32_t rounds = s_rounds;
    for (uint32_t round = 0; round < rounds; ++round) {
        int ps[2];
        ps[0] = int(round) - 2 * int(wave);
        ps[1] = ps[0] - 1;
        bool vh[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) vh[h] = live && ps[h] >= 0 && uint32_t(ps[h]) < nstrips && nblk[h] > 0;
        if (vh[0] || vh[1]) {
            uint32_t jb[2];
            bool has_cols[2], wr_col[2];
            int lastc[2];
            int2* colout[2];
            const int2* colin[2];
            const uint32_t* profb[2];  // this lane's 4 words of profile row 0, per half
            int base[2];               // what relative 0 stands for (wave-uniform)
            uint32_t ncl[2];           // lanes that own columns of the strip
            v2s Hp[FW_C], F[FW_C];
            v2s dg;
            {
                int Ha[2][FW_C], Fa[2][FW_C], dga[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t p = uint32_t(vh[h] ? ps[h] : 0);
                    jb[h] = p * strip_cols + lane * FW_C;
                    has_cols[h] = vh[h] && jb[h] < m;
                    ncl[h] = vh[h] ? min(64u, (m - p * strip_cols + FW_C - 1) / FW_C) : 0u;
                    lastc[h] = (vh[h] && m - 1 >= jb[h] && m - 1 < jb[h] + FW_C) ? int(m - 1 - jb[h]) : -1;
                    const uint32_t jr = jb[h] + FW_C;
                    wr_col[h] = vh[h] && (jr % TILE) == 0 && jr < m;
                    colout[h] = wr_col[h] ? colck + uint64_t(jr / TILE - 1) * col_pitch(n) : colck;
                    colin[h] = p ? colck + uint64_t(p * strip_cols / TILE - 1) * col_pitch(n) : colck;
                    profb[h] = &s_prof_all[wv][p & 1u][0][lane][0];
                    if (!vh[h]) {
#pragma unroll
                        for (int c = 0; c < FW_C; ++c) {
                            Ha[h][c] = 0;
                            Fa[h][c] = 0;
                        }
                        dga[h] = 0;
                    } else if (r_lo[h] == 0) {
                        const int b0 = ge * int(jb[h]) - gd;
#pragma unroll
                        for (int c = 0; c < FW_C; ++c) {
                            Ha[h][c] = b0 + ge * (c + 1);
                            Fa[h][c] = ALN_NEG;
                        }
                        dga[h] = b0;
                    } else {
                        // the last row of the band above: band - 1 = (wave w', half h') wrote entry w' * tpb + tpb - 1
                        const uint32_t bp = 2u * wave + uint32_t(h) - 1u;
                        const uint32_t* ent = reinterpret_cast<const uint32_t*>(rowck) + uint64_t((bp >> 1) * tpb + tpb - 1u) * row16_stride(m);
                        const int hs = int(bp & 1u);
#pragma unroll
                        for (int c = 0; c < FW_C; ++c) {
                            int2 v{0, 0};
                            if (jb[h] + c < m) v = int2{row16_get(ent, m, hs, 0, jb[h] + c), row16_get(ent, m, hs, 1, jb[h] + c)};
                            Ha[h][c] = v.x;
                            Fa[h][c] = v.y;
                        }
                        dga[h] = (jb[h] > 0 && jb[h] <= m) ? row16_get(ent, m, hs, 0, jb[h] - 1) : ge * int(r_lo[h]) - gd;  // column 0 holds H = 0
                    }
                    base[h] = __builtin_amdgcn_readfirstlane(Ha[h][0]);  // lane 0, first column of the strip
                }
#pragma unroll
                for (int c = 0; c < FW_C; ++c) {
                    // (columns right of the matrix: relative 0, they only ever follow their left neighbours)
                    const bool in0 = jb[0] + c < m, in1 = jb[1] + c < m;
                    Hp[c] = pack2(in0 ? clamp16(Ha[0][c] - base[0]) : 0, in1 ? clamp16(Ha[1][c] - base[1]) : 0);
                    F[c] = pack2(in0 ? clamp16(Fa[0][c] - base[0]) : 0, in1 ? clamp16(Fa[1][c] - base[1]) : 0);
                }
                dg = pack2(clamp16(dga[0] - base[0]), clamp16(dga[1] - base[1]));
            }
            // the low half's strip is new: its profile goes to buffer (strip & 1); the high half's strip was
            // the low half's one round ago
            if (vh[0]) {
                uint32_t rpk[FW_C / 4];
#pragma unroll
                for (int c4 = 0; c4 < FW_C / 4; ++c4) {
                    uint32_t w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w |= ref_byte(r, m, pr.rc, jb[0] + c4 * 4 + e) << (8 * e);
                    rpk[c4] = w;
                }
                uint32_t(*pf)[64][4] = s_prof_all[wv][uint32_t(ps[0]) & 1u];
                const uint32_t bases = 'A' | ('C' << 8) | ('G' << 16) | ('T' << 24);
#pragma unroll
                for (int code = 0; code < 5; ++code) {
                    const uint32_t b = code < 4 ? (bases >> (8 * code)) & 0xFFu : 0x100u;  // 0x100: equals no byte
#pragma unroll
                    for (int c4 = 0; c4 < FW_C / 4; ++c4) {
                        uint32_t w = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w |= uint32_t(((rpk[c4] >> (8 * e)) & 0xFFu) == b ? cm : cx) << (8 * e);
                        pf[code][lane][c4] = w;
                    }
                }
            }
            // left edge of a strip (absolute, slanted): column 0 of the matrix or the column checkpoint
            auto left_edge = [&](int h, uint32_t row) {
                int2 v{ge * int(row + 1) - gd, ALN_NEG};
                if (vh[h] && ps[h] > 0 && row < r_hi[h]) v = colin[h][row];
                return v;
            };
            uint32_t qn = 0;
            int2 en[2] = {{0, 0}, {0, 0}};
            v2s hl[FW_R], el[FW_R];
            uint32_t qc[FW_R];
#pragma unroll
            for (int rr = 0; rr < FW_R; ++rr) {
                hl[rr] = pack2(0, 0);
                el[rr] = pack2(-32768, -32768);
                qc[rr] = 0;
            }
            auto publish = [&](uint32_t blk) {  // relative to the bases of NOW (the rebase keeps the ring current)
                s_look[0][(blk & 1u) * 64u + lane] = as_u32(pack2(clamp16(en[0].x - base[0]), clamp16(en[1].x - base[1])));
                s_look[1][(blk & 1u) * 64u + lane] = as_u32(pack2(clamp16(en[0].y - base[0]), clamp16(en[1].y - base[1])));
                s_look[2][(blk & 1u) * 64u + lane] = qn;
            };
            auto fetch = [&](uint32_t blk) {
                uint32_t qq[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t row = r_lo[h] + blk * 64u + lane;
                    qq[h] = qcode((vh[h] && row < r_hi[h]) ? q[row] : 0u);
                    en[h] = left_edge(h, row);
                }
                qn = qq[0] | (qq[1] << 16);
            };
            fetch(0);
            publish(0);
            fetch(1);
            uint4 nxh, nxe, nxq;
            {
                nxh = *reinterpret_cast<const uint4*>(&s_look[0][0]);
                nxe = *reinterpret_cast<const uint4*>(&s_look[1][0]);
                nxq = *reinterpret_cast<const uint4*>(&s_look[2][0]);
            }
            for (uint32_t s = 0; s < nsteps; ++s) {
                if ((s & (64 / FW_R - 1)) == 0) {
                    if (s) {
                        // ---- rebase: advance each base by the first column of a lane that is inside its band ----
                        int dl[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            // (the lowest lane still inside the band; lanes right of the matrix own no columns)
                            const uint32_t lr = s + 1u > nblk[h] ? min(63u, s + 1u - nblk[h]) : 0u;
                            const int ref = __builtin_amdgcn_readlane(int(as_u32(Hp[0])), int(lr));
                            dl[h] = max(half_of_rt(ref, h), 0);
                            if (!vh[h] || lr >= ncl[h]) dl[h] = 0;
                            base[h] += dl[h];
                        }
                        const v2s d2 = pack2(dl[0], dl[1]);
                        // guard (lanes inside a band only: the registers of the others are idle)
                        {
                            v2s mx = Hp[0], mn = Hp[0];
#pragma unroll
                            for (int c = 1; c < FW_C; ++c) {
                                mx = pmax(mx, Hp[c]);
                                mn = pmin(mn, Hp[c]);
                            }
                            const int bi = int(s) - int(lane);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int hi_v = h ? int(mx.y) : int(mx.x), lo_v = h ? int(mn.y) : int(mn.x);
                                if (has_cols[h] && bi >= 0 && uint32_t(bi) < nblk[h] && (hi_v > P16_GUARD || lo_v < -P16_GUARD)) ovf = 1;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < FW_C; ++c) {
                            Hp[c] = psubs(Hp[c], d2);
                            F[c] = psubs(F[c], d2);
                        }
                        dg = psubs(dg, d2);
#pragma unroll
                        for (int rr = 0; rr < FW_R; ++rr) {
                            hl[rr] = psubs(hl[rr], d2);
                            el[rr] = psubs(el[rr], d2);
                        }
                        nxh = uint4{as_u32(psubs(as_v2s(nxh.x), d2)), as_u32(psubs(as_v2s(nxh.y), d2)), as_u32(psubs(as_v2s(nxh.z), d2)),
                                    as_u32(psubs(as_v2s(nxh.w), d2))};
                        nxe = uint4{as_u32(psubs(as_v2s(nxe.x), d2)), as_u32(psubs(as_v2s(nxe.y), d2)), as_u32(psubs(as_v2s(nxe.z), d2)),
                                    as_u32(psubs(as_v2s(nxe.w), d2))};
                        // the block of the ring that is being consumed
                        const uint32_t cur = ((s / (64 / FW_R)) & 1u) * 64u + lane;
                        s_look[0][cur] = as_u32(psubs(as_v2s(s_look[0][cur]), d2));
                        s_look[1][cur] = as_u32(psubs(as_v2s(s_look[1][cur]), d2));
                    }
                    const uint32_t blk = s / (64 / FW_R) + 1u;
                    publish(blk);
                    fetch(blk + 1u);
                }
                {
                    const uint32_t h4[FW_R] = {nxh.x, nxh.y, nxh.z, nxh.w}, e4[FW_R] = {nxe.x, nxe.y, nxe.z, nxe.w},
                                   q4[FW_R] = {nxq.x, nxq.y, nxq.z, nxq.w};
#pragma unroll
                    for (int rr = 0; rr < FW_R; ++rr) {
                        hl[rr] = dpp_left_or(hl[rr], as_v2s(h4[rr]));
                        el[rr] = dpp_left_or(el[rr], as_v2s(e4[rr]));
                        qc[rr] = from_left_or(qc[rr], q4[rr]);
                    }
                    const uint32_t nx = ((s + 1u) * FW_R) & 127u;
                    nxh = *reinterpret_cast<const uint4*>(&s_look[0][nx]);
                    nxe = *reinterpret_cast<const uint4*>(&s_look[1][nx]);
                    nxq = *reinterpret_cast<const uint4*>(&s_look[2][nx]);
                }
                const int bi = int(s) - int(lane);
                bool act[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) act[h] = has_cols[h] && bi >= 0 && uint32_t(bi) < nblk[h];
                if (act[0] || act[1]) {
                    const bool special = (act[0] && (lastc[0] >= 0 || (last_band[0] && uint32_t(bi) + 1u == nblk[0]))) ||
                                         (act[1] && (lastc[1] >= 0 || (last_band[1] && uint32_t(bi) + 1u == nblk[1])));
                    auto row_step = [&](int rr) {
                        const v2s hl_in = hl[rr];
                        const uint4 xa = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(profb[0]) + (qc[rr] & 0xFFFFu));
                        const uint4 xb = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(profb[1]) + (qc[rr] >> 16));
                        const uint32_t xl[FW_C / 4] = {xa.x, xa.y, xa.z, xa.w}, xh[FW_C / 4] = {xb.x, xb.y, xb.z, xb.w};
                        fwd16_cells(Hp, F, xl, xh, hl[rr], el[rr], dg, gd2);
                        dg = hl_in;
                    };
                    if (!special) {
#pragma unroll
                        for (int rr = 0; rr < FW_R; ++rr) row_step(rr);
                    } else {
#pragma unroll
                        for (int rr = 0; rr < FW_R; ++rr) {
                            row_step(rr);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const uint32_t i = r_lo[h] + uint32_t(bi) * FW_R + rr;  // 0-based row
                                if (act[h] && i < r_hi[h]) {
                                    if (lastc[h] >= 0) {
                                        const uint32_t lc = opaque(uint32_t(lastc[h]));
                                        const bool b0 = lc & 1u, b1 = lc & 2u, b2 = lc & 4u, b3 = lc & 8u;
                                        const uint32_t s0 = b0 ? as_u32(Hp[1]) : as_u32(Hp[0]), s1 = b0 ? as_u32(Hp[3]) : as_u32(Hp[2]),
                                                       s2 = b0 ? as_u32(Hp[5]) : as_u32(Hp[4]), s3 = b0 ? as_u32(Hp[7]) : as_u32(Hp[6]),
                                                       s4 = b0 ? as_u32(Hp[9]) : as_u32(Hp[8]), s5 = b0 ? as_u32(Hp[11]) : as_u32(Hp[10]),
                                                       s6 = b0 ? as_u32(Hp[13]) : as_u32(Hp[12]), s7 = b0 ? as_u32(Hp[15]) : as_u32(Hp[14]);
                                        const uint32_t u0 = b1 ? s1 : s0, u1 = b1 ? s3 : s2, u2 = b1 ? s5 : s4, u3 = b1 ? s7 : s6;
                                        const uint32_t v0 = b2 ? u1 : u0, v1 = b2 ? u3 : u2;
                                        int hm = half_of_rt(int(b3 ? v1 : v0), h) + base[h];
                                        hm += gd - ge * int(opaque(i) + 1 + m);  // true H(i + 1, m)
                                        if (hm > bc[h]) {
                                            bc[h] = hm;
                                            bc_i[h] = i + 1u;
                                        }
                                    }
                                    if (i + 1u == n) {  // last row: this lane's best cell, first column wins ties
                                        int br = ALN_NEG;
                                        uint32_t bj = 0;
                                        const uint32_t jbo = opaque(jb[h]);
                                        const int b0 = base[h] + gd - ge * int(n + jbo);
#pragma unroll
                                        for (int c = 0; c < FW_C; ++c) {
                                            const int ht = (h ? int(Hp[c].y) : int(Hp[c].x)) + b0 - ge * (c + 1);  // true H(n, jb + c + 1)
                                            if (jbo + c < m && ht > br) {
                                                br = ht;
                                                bj = jbo + c + 1;
                                            }
                                        }
                                        mylrow[jbo / FW_C] = int2{br, int(bj)};
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t i0 = r_lo[h] + uint32_t(bi) * FW_R;
                        if (act[h] && wr_col[h]) {
                            int ha[FW_R], ea[FW_R];
#pragma unroll
                            for (int rr = 0; rr < FW_R; ++rr) {
                                ha[rr] = (h ? int(hl[rr].y) : int(hl[rr].x)) + base[h];
                                ea[rr] = (h ? int(el[rr].y) : int(el[rr].x)) + base[h];
                            }
                            if (!special) {  // i0 is a multiple of 4: two 16-byte stores
                                int4* co = reinterpret_cast<int4*>(colout[h] + i0);
                                co[0] = int4{ha[0], ea[0], ha[1], ea[1]};
                                co[1] = int4{ha[2], ea[2], ha[3], ea[3]};
                            } else {
#pragma unroll
                                for (int rr = 0; rr < FW_R; ++rr)
                                    if (i0 + rr < r_hi[h]) colout[h][i0 + rr] = int2{ha[rr], ea[rr]};
                            }
                        }
                    }
                    // row checkpoint: the block's last row closes a tile (in both bands at once: bands start on tile
                    // boundaries).  The registers go out as they are, with the bases of the moment.
                    {
                        const uint32_t i1 = uint32_t(bi + 1) * FW_R;  // rows of the band that are done
                        if ((i1 % TILE) == 0) {
                            uint32_t* ent = reinterpret_cast<uint32_t*>(rowck) + uint64_t(wave * tpb + (i1 / TILE - 1u)) * row16_stride(m);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (act[h] && r_lo[h] + i1 < n) {
                                    const uint32_t jbo = opaque(jb[h]);
                                    uint4* oh = reinterpret_cast<uint4*>(ent + uint64_t(2 * h) * row_pitch(m) + jbo);
                                    uint4* of = reinterpret_cast<uint4*>(ent + uint64_t(2 * h + 1) * row_pitch(m) + jbo);
#pragma unroll
                                    for (int c = 0; c < FW_C; c += 4) {
                                        oh[c / 4] = uint4{as_u32(Hp[c]), as_u32(Hp[c + 1]), as_u32(Hp[c + 2]), as_u32(Hp[c + 3])};
                                        of[c / 4] = uint4{as_u32(F[c]), as_u32(F[c + 1]), as_u32(F[c + 2]), as_u32(F[c + 3])};
                                    }
                                    ent[4u * row_pitch(m) + uint32_t(h) * uint32_t(row_pitch(m) / 16u) + jbo / FW_C] = uint32_t(base[h]);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (lastc[h] >= 0) {
                    s_best[2u * wv + uint32_t(h)][0] = bc[h];
                    s_best[2u * wv + uint32_t(h)][1] = int(bc_i[h]);
                }
        }
        __syncthreads();  // orders this round's checkpoints before the next round reads them
    }
    if (ovf) s_ovf[wv] = 1;
    __syncthreads();

    // end cell: best of the last column (rows ascending), replaced only by a strictly larger cell of the
    // last row (columns ascending from 0) — the host aligner's scan order (ioc_align.cpp)
    if (live && wave == 0) {
        int br = 0;  // H(n, 0)
        uint32_t bj = 0;
        for (uint32_t e = lane; e < nstrips * 64u; e += 64) {
            if (uint64_t(e) * FW_C >= m) continue;
            const int2 x = mylrow[e];
            if (x.x > br || (x.x == br && uint32_t(x.y) < bj)) {
                br = x.x;
                bj = uint32_t(x.y);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int s = __shfl_xor(br, o);
            const uint32_t j = __shfl_xor(bj, o);
            if (s > br || (s == br && j < bj)) {
                br = s;
                bj = j;
            }
        }
        if (lane == 0) {
            int fin = ALN_NEG;
            uint32_t bi = 0, bjj = m;
            uint32_t bad = 0;
            for (uint32_t w2 = 0; w2 < nwaves; ++w2) bad |= s_ovf[slot * wpp + w2];
            for (uint32_t b2 = 0; b2 < nbands; ++b2) {  // bands top to bottom: the first row wins ties
                if (min(n, b2 * tpb * TILE) >= n) break;
                if (s_best[2u * slot * wpp + b2][0] > fin) {
                    fin = s_best[2u * slot * wpp + b2][0];
                    bi = uint32_t(s_best[2u * slot * wpp + b2][1]);
                }
            }
            if (br > fin) {
                fin = br;
                bi = n;
                bjj = bj;
            }
            ends[pid] = int4{fin, int(bi), int(bjj), int(bad | 2u | (nbands << 8))};  // [0] refused  [1] packed row checkpoints  [15:8] bands
        }
    }
}

// sliding k-window counter over the comparison string, fed in REVERSE order: getAlnRatio counts the
// windows [i, i + k) for i = 0 .. len - k - 1, i.e. every window but the last one — in reverse order,
// every window but the first one to complete.
struct WinStat {
    uint32_t wb = 0, pushed = 0, cnt = 0;
    __device__ __forceinline__ void push(uint32_t bit, uint32_t kmask, uint32_t k, int il)
    {
        wb = ((wb << 1) | bit) & kmask;
        ++pushed;
        if (pushed > k && int(__popc(wb)) >= il) ++cnt;
    }
    // m <= 64 pushes at once, bit x of `bits` = the x-th character pushed; every lane gets the same arguments
    // and lane x evaluates the window as it stands after push x
    __device__ __forceinline__ void push_run(unsigned long long bits, uint32_t m, uint32_t kmask, uint32_t k, int il, uint32_t lane)
    {
        // the characters up to push `lane`, the latest in bit 0 (as the window keeps them)
        const unsigned long long upto = __brevll(bits << (63u - lane));
        const unsigned long long old = lane + 1u < 64u ? (unsigned long long)wb << (lane + 1u) : 0ull;
        const uint32_t win = uint32_t(old | upto) & kmask;
        const bool hit = lane < m && pushed + lane + 1u > k && int(__popc(win)) >= il;
        cnt += uint32_t(__popcll(__ballot(hit)));
        wb = uint32_t(__builtin_amdgcn_readlane(int(win), int(m - 1u)));
        pushed += m;
    }
    __device__ void blanks(uint32_t g, uint32_t kmask, uint32_t k, int il)
    {
        const uint32_t t = g < k ? g : k;
        for (uint32_t x = 0; x < t; ++x) push(0u, kmask, k, il);
        if (g > t) {  // the window is all blanks from here on
            const uint32_t rest = g - t;
            const uint32_t first_counted = pushed >= k ? 0u : k - pushed;  // pushes until pushed > k holds
            pushed += rest;
            if (il <= 0 && rest > first_counted) cnt += rest - first_counted;
        }
    }
};

// Pass 2: one wave per pair.  Direction nibble of a cell: bits 0-1 = where H came from (0 diagonal with
// identical bases, 3 diagonal with different bases, 1 E, 2 F), bit 2 = E extended, bit 3 = F extended.
constexpr int TR_WAVES = 4;
// what a barrier is to a one-wave workgroup: LDS operations of a wave complete in order, the compiler must not move them
__device__ __forceinline__ void tr_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void __launch_bounds__(64 * TR_WAVES)
k_align_trace(const AlnPairDev* __restrict__ pairs, const uint32_t* __restrict__ order, const uint8_t* __restrict__ pool,
              AlnParams P, const int2* __restrict__ ck, const AlnCk* __restrict__ cko, const int4* __restrict__ ends,
              int32_t* __restrict__ out_score, uint32_t* __restrict__ out_count, uint32_t count)
{
    // A workgroup is TR_WAVES = 4 independent waves, one pair each, with LDS of their own: a workgroup of four waves puts
    // one on every SIMD of its CU, whereas one-wave workgroups were seen three to a SIMD on some CUs (each of them then
    // half as fast: 14.5 against 7.6 ms) while other SIMDs held one.  No workgroup barrier anywhere: a wave only ever
    // reads what it wrote itself, in program order.
    __shared__ uint16_t dirs_all[TR_WAVES][TILE * TR_C / 4][64];  // TR_C nibbles per lane and row (TILE 128: used as bytes [row][lane])
    __shared__ int2 s_left_all[TR_WAVES][TILE];
    __shared__ uint32_t s_q_all[TR_WAVES][TILE];
    const uint32_t wv = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));  // uniform: the pair's state stays in SGPRs
    const uint32_t pslot = blockIdx.x * TR_WAVES + wv;
    if (pslot >= count) return;
    uint16_t(*dirs)[64] = dirs_all[wv];
    int2* s_left = s_left_all[wv];
    uint32_t* s_q = s_q_all[wv];
    const uint32_t pid = order[pslot];
    const AlnPairDev pr = pairs[pid];
    const uint32_t n = pr.n, m = pr.m;
    const int go = pr.gap_open, il = pr.ilimit;
    const uint8_t* __restrict__ q = pool + pr.q_off;
    const uint8_t* __restrict__ r = pool + pr.r_off;
    const int2* __restrict__ rowck = ck + cko[pid].row_off;
    const int2* __restrict__ colck = ck + cko[pid].col_off;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = P.k, kmask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const int4 en = ends[pid];
    if (en.w & 1) {  // the packed forward pass left its 16-bit window: the host sends the pair to the 32-bit kernel
        if (lane == 0) {
            out_score[pid] = INT32_MIN;
            out_count[pid] = 0xFFFFFFFFu;
        }
        return;
    }
    uint32_t i = uint32_t(en.y), j = uint32_t(en.z);
    int state = 0;  // 0 = H, 1 = E, 2 = F
    WinStat ws;
    ws.blanks((m - j) + (n - i), kmask, k, il);  // trailing end gaps are the tail of the string

#ifdef TR_PROF
    unsigned long long tp_load = 0, tp_comp = 0, tp_walk = 0, tp_t = __builtin_readcyclecounter();
    uint32_t tp_tiles = 0;
#endif
    while (i > 0 && j > 0) {
        const uint32_t r0 = ((i - 1) / TILE) * TILE, c0 = ((j - 1) / TILE) * TILE;
        const uint32_t rows = i - r0, cols = j - c0;
        // The tile is recomputed on the forward pass's own SLANTED scores (see fwd_cells: X* = X + ge (i + j), Hq = H* - gd),
        // exactly as the checkpoints hold them: a gap extension costs nothing, an opening is already inside Hq, so a cell is
        // two max + max3 + add + sub, and every decision of the host aligner's cell is a comparison of the same operands
        // (E extended <=> E* > Hq of the left cell; H from the diagonal <=> H* equals it, which wins ties, else from E if
        // equal to E*, else from F) — 18 VALU per cell where the unslanted form took 24.  This kernel is bound by VALU
        // issue on the SIMDs that hold two of its waves.
        const int gd = go - P.gap_extend, ge = P.gap_extend;
        int cm = P.match + 2 * ge + gd, cx = P.mismatch + 2 * ge + gd;
        asm volatile("" : "+v"(cm), "+v"(cx));  // kept in VGPRs: the select below cannot take two scalars, and the compiler would copy them over in every step
        for (uint32_t x = lane; x < rows; x += 64) {
            s_q[x] = q[r0 + x];
            int2 le{ge * int(r0 + x + 1) - gd, ALN_NEG};  // column 0: H = 0, no gap to extend
            if (c0) le = colck[uint64_t(c0 / TILE - 1) * col_pitch(n) + r0 + x];  // (Hq, E*) as passed between lanes
            s_left[x] = le;
        }
        const uint32_t jb = c0 + lane * TR_C;  // columns to the left of this lane's block
        uint32_t rpk[1];
        {
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < TR_C; ++e) w |= ref_byte(r, m, pr.rc, jb + e) << (8 * e);
            rpk[0] = w;
        }
        int Hp[TR_C], F[TR_C];  // (Hq, F*) of the row above
        int dg = ge * int(r0 + jb) - gd;  // Hq(r0, jb) where H = 0: row 0, or column 0
        if (r0 == 0) {
#pragma unroll
            for (int c = 0; c < TR_C; ++c) {
                Hp[c] = ge * int(jb + c + 1) - gd;
                F[c] = ALN_NEG;
            }
        } else if (en.w & 2) {  // written by k_align_fwd16: 16-bit values + bases, by (wave, tile of the band)
            const uint32_t nb = uint32_t(en.w) >> 8, tpb = ((n + TILE - 1) / TILE + nb - 1) / nb;
            const uint32_t band = (r0 / TILE - 1) / tpb, kt = (r0 / TILE - 1) % tpb;
            const uint32_t* ent = reinterpret_cast<const uint32_t*>(rowck) + uint64_t((band >> 1) * tpb + kt) * row16_stride(m);
            const int hs = int(band & 1u);
#pragma unroll
            for (int c = 0; c < TR_C; ++c) {
                int2 v{0, ALN_NEG};
                if (jb + c < m) v = int2{row16_get(ent, m, hs, 0, jb + c), row16_get(ent, m, hs, 1, jb + c)};
                Hp[c] = v.x;
                F[c] = v.y;
            }
            if (jb > 0 && jb <= m) dg = row16_get(ent, m, hs, 0, jb - 1);
        } else {
            const int* roh = reinterpret_cast<const int*>(rowck + uint64_t(r0 / TILE - 1) * row_pitch(m));
            const int* rof = roh + row_pitch(m);
#pragma unroll
            for (int c = 0; c < TR_C; ++c) {
                int2 v{0, ALN_NEG};
                if (jb + c < m) v = int2{roh[jb + c], rof[jb + c]};
                Hp[c] = v.x;
                F[c] = v.y;
            }
            if (jb > 0 && jb <= m) dg = roh[jb - 1];
        }
        tr_wave_sync();
#ifdef TR_PROF
        { const unsigned long long t = __builtin_readcyclecounter(); tp_load += t - tp_t; tp_t = t; ++tp_tiles; }
#endif
        const uint32_t nact = (cols + TR_C - 1) / TR_C;
        const uint32_t nsteps = rows + nact - 1;
        int out_h = 0, out_e = ALN_NEG;
        uint32_t out_q = 0;
        // lane 0's inputs of a step (left edge and query byte of row s) are read one step AHEAD, by every lane at one address,
        // and enter the wave as the `old` operand of the shifts
        int2 nle = s_left[0];
        uint32_t nq = s_q[0];
        auto tstep = [&](const uint32_t s) __attribute__((always_inline)) {
            int hl = int(from_left_or(uint32_t(out_h), uint32_t(nle.x)));
            int el = int(from_left_or(uint32_t(out_e), uint32_t(nle.y)));
            uint32_t qc = from_left_or(out_q, nq);
            {
                const uint32_t sn = s + 1u < rows ? s + 1u : rows - 1u;  // (rows past the end are never looked at)
                nle = s_left[sn];
                nq = s_q[sn];
            }
            const int ri = int(s) - int(lane);
            if (ri >= 0 && uint32_t(ri) < rows && lane < nact) {
                const int hl_in = hl;
                uint32_t bits = 0;
#pragma unroll
                for (int c = 0; c < TR_C; ++c) {
                    const bool ex = el > hl;
                    const int E = max(el, hl);
                    const bool fx = F[c] > Hp[c];
                    const int Fn = max(F[c], Hp[c]);
                    const bool mt = qc == ((rpk[0] >> (8 * c)) & 0xFFu);
                    const int hd = dg + (mt ? cm : cx);
                    const int h = max(max(hd, E), Fn);
                    // the host aligner's cell (ioc_align.cpp): H = diagonal, replaced by E if E > H, then by F if F > H
                    const uint32_t from = h == hd ? (mt ? 0u : 3u << (4 * c)) : (h == E ? 1u << (4 * c) : 2u << (4 * c));
                    bits |= from | (ex ? 4u << (4 * c) : 0u) | (fx ? 8u << (4 * c) : 0u);
                    dg = Hp[c];
                    Hp[c] = h - gd;
                    F[c] = Fn;
                    hl = h - gd;
                    el = E;
                }
                dg = hl_in;
                if (TR_C == 4)
                    dirs[ri][lane] = uint16_t(bits);
                else
                    reinterpret_cast<uint8_t*>(&dirs[0][0])[uint32_t(ri) * 64u + lane] = uint8_t(bits);  // (TILE 128: a byte per lane and row, row-major)
            }
            out_h = hl;
            out_e = el;
            out_q = qc;
        };
        {   // two steps per iteration (the compiler does not unroll a loop with wave-level operations): fewer register copies
            uint32_t s = 0;
#if IOC_TR_STEP_UNROLL == 4
            for (; s + 3u < nsteps; s += 4) {
                tstep(s);
                tstep(s + 1u);
                tstep(s + 2u);
                tstep(s + 3u);
            }
#endif
            for (; s + 1u < nsteps; s += 2) {
                tstep(s);
                tstep(s + 1u);
            }
            if (s < nsteps) tstep(s);
        }
        tr_wave_sync();
#ifdef TR_PROF
        { const unsigned long long t = __builtin_readcyclecounter(); tp_comp += t - tp_t; tp_t = t; }
#endif
        // the host aligner's traceback loop inside this tile (identical in every lane)
        while (i > r0 && j > c0) {
            const uint32_t cj = j - c0 - 1;
            const uint32_t ri = i - r0 - 1;
            const uint32_t t = TR_C == 4 ? (uint32_t(dirs[ri][cj / TR_C]) >> (4 * (cj % TR_C))) & 0xFu
                                         : (uint32_t(reinterpret_cast<const uint8_t*>(&dirs[0][0])[ri * 64u + cj / TR_C]) >> (4 * (cj % TR_C))) & 0xFu;
            if (state == 0) {
                // look ahead along the diagonal: lane l reads the cell l steps up-left; the run of diagonal moves
                // from here on goes into the window counter at once
                const bool inr = i - r0 > lane && j - c0 > lane;
                uint32_t tl = 1u;
                if (inr) {
                    const uint32_t rl = ri - lane, cl = cj - lane;
                    tl = (TR_C == 4 ? (uint32_t(dirs[rl][cl / TR_C]) >> (4 * (cl % TR_C)))
                                    : (uint32_t(reinterpret_cast<const uint8_t*>(&dirs[0][0])[rl * 64u + cl / TR_C]) >> (4 * (cl % TR_C)))) & 3u;
                }
                const bool dgl = inr && (tl == 0u || tl == 3u);
                const unsigned long long dm = __ballot(dgl);
                const uint32_t run = ~dm ? uint32_t(__builtin_ctzll(~dm)) : 64u;
                if (run > 0) {
                    const unsigned long long mb = __ballot(dgl && tl == 0u) & (run == 64u ? ~0ull : ((1ull << run) - 1ull));
                    ws.push_run(mb, run, kmask, k, il, lane);
                    i -= run;
                    j -= run;
                } else {
                    state = (t & 3u) == 1u ? 1 : 2;
                }
            } else if (state == 1) {
                ws.push(0u, kmask, k, il);
                if (!(t & 4u)) state = 0;
                --j;
            } else {
                ws.push(0u, kmask, k, il);
                if (!(t & 8u)) state = 0;
                --i;
            }
        }
        tr_wave_sync();
#ifdef TR_PROF
        { const unsigned long long t = __builtin_readcyclecounter(); tp_walk += t - tp_t; tp_t = t; }
#endif
    }
    ws.blanks(i + j, kmask, k, il);  // leading end gaps
    if (lane == 0) {
        out_score[pid] = en.x;
        out_count[pid] = ws.cnt;
#ifdef TR_PROF
        if (pslot == 0) printf("trace profile (cycles of the 100 MHz counter x tiles %u): load %llu, recompute %llu, walk %llu\n", tp_tiles, tp_load, tp_comp, tp_walk);
#endif
    }
}

// per sequence: does it hold a byte other than A C G T?  (The query-profile kernel knows four bases; a pair
// with anything else — the host aligner matches any two equal bytes — takes the comparing kernel.)
__global__ void __launch_bounds__(256)
k_seq_flags(const uint8_t* __restrict__ pool, const int64_t* __restrict__ offs, uint8_t* __restrict__ flags)
{
    __shared__ uint32_t s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    const int64_t a = offs[blockIdx.x], b = offs[blockIdx.x + 1];
    uint32_t any = 0;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) {
        const uint8_t ch = pool[i];
        any |= (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ? 1u : 0u;
    }
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = uint8_t(s_any);
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
    ioc_poison(b.p, want);
    return IOC_OK;
}

#define ACHK(c, call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

#include "ioc_align_v2.inc"

// ---- host side of version 2 -------------------------------------------------------------------------------------------
// the pairs order[0 .. cnt) (all of the query-profile kind, heaviest first) through k_fwd2 / k_fwd2_ends / k_trace2.
// Returns IOC_OK with *fell_back = true when a bounded wait of the forward pass ran out (the caller then runs the same
// pairs through version 1).
// set while pairs the corridor could not vouch for are re-run: every tile of their grids
thread_local bool g_no_corridor = false;

int align_v2_run(ioc_ctx* c, const std::vector<AlnPairDev>& dp, const uint32_t* order, uint32_t cnt, const uint32_t* d_order,
                 const AlnParams& P, int32_t* d_score, uint32_t* d_count, bool* fell_back)
{
    *fell_back = false;
    if (cnt == 0) return IOC_OK;
    hipStream_t s = c->stream;
    int r;
    const auto t_plan0 = std::chrono::steady_clock::now();  // (IOC_TRACE: host time of the planning up to the first launch)
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    if (n_cu < 1) n_cu = 256;
    size_t free_b = 0, total_b = 0;
    ACHK(c, hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = uint64_t(free_b + c->a_ck.cap) / 2;
    if (const char* e = getenv("IOC_ALIGN_CK_BUDGET_MB")) budget = uint64_t(atoll(e)) << 20;
    // the 16-bit window's guard (|relative score| at a rebase): a pair beyond it is flagged and re-run by version 1
    int guard = P16_GUARD;
    if (const char* e = getenv("IOC_ALIGN_V2_GUARD")) guard = std::max(1, std::min(P16_GUARD, atoi(e)));
    const uint32_t np = uint32_t(dp.size());
    const uint32_t ncouples = (cnt + 1u) / 2u;
    uint32_t probe_rows = 1024;
    if (const char* e = getenv("IOC_ALIGN_PROBE_ROWS")) probe_rows = uint32_t(std::max(512, atoi(e)));
    std::vector<int32_t> pend_cert(np, INT32_MIN);
    uint64_t tiles_skipped = 0, tiles_all = 0;
    // The corridor's half width as a fraction of the longer sequence (V2Couple; IOC_ALIGN_CORRIDOR=0: every tile; a pair whose
    // result the corridor cannot vouch for is run again with g_no_corridor set).  0.2: a pair of one transcript scores 1.62 - 1.8
    // per base at match 2 — it passes if 2 (1 - B / len) < 1.62 — and the tile grid of a 16.7 kb pair shrinks to about half.
    double corridor_frac = g_no_corridor ? 0.0 : 0.2;
    bool corridor_fixed = false;  // (IOC_ALIGN_CORRIDOR: one fraction for every couple, as in round 4)
    if (const char* e = getenv("IOC_ALIGN_CORRIDOR")) {
        corridor_frac = g_no_corridor ? 0.0 : std::max(0.0, std::min(1.0, atof(e)));
        corridor_fixed = true;
    }
    // ROUND 5: the half width PER COUPLE, from the pairs' summed error rates.  A pair of one transcript scores
    // rho = match - e x c per base, and passes the certificate iff B / length > 1 - rho / match.  Before anything has been aligned
    // c comes from the error model behind setGapOpen's classes — a third each of substitutions (match - mismatch lost), insertions
    // and deletions (gap open + the match) — plus 0.02 x length; once a gap-open class has 64 related pairs behind it (this
    // context's earlier batches: ioc_ctx::aln_fit, fed in ioc_align_pairs) the line fitted to THEM, 4.5 standard deviations down.
    // Config 3: mean width 0.157 x length (round 4: 0.2 for everybody — set by the least similar pair of the batch).  A wrong
    // guess costs time (the probe widens, or the pair is refuted and run again), never a result.
    const bool fit_ok = c->aln_fit_sig[0] == P.match && c->aln_fit_sig[1] == P.mismatch && c->aln_fit_sig[2] == P.gap_extend && !getenv("IOC_ALIGN_CORRIDOR_NO_FIT");
    auto pair_frac = [&](const AlnPairDev& d) -> double {
        if (corridor_fixed || !(P.match > 0)) return corridor_frac;
        const double e = double(d.e_sum);
        if (!(e > 0.0) || !(e < 1.0)) return corridor_frac;
        const int go = std::max(2, std::min(5, d.gap_open));
        double rho = double(P.match) - e * (double(P.match - P.mismatch) / 3.0 + 2.0 / 3.0 * double(go + P.match)), margin = 0.02;
        const ioc_ctx::CorridorFit& f = c->aln_fit[go - 2];
        if (fit_ok && f.n >= 64.0 && e >= f.e_lo - 0.02 && e <= f.e_hi + 0.02) {
            const double me = f.se / f.n, mr = f.sr / f.n, vee = f.see / f.n - me * me, ver = f.ser / f.n - me * mr, vrr = f.srr / f.n - mr * mr;
            const double slope = vee > 1e-9 ? ver / vee : 0.0;
            const double sd = std::sqrt(std::max(0.0, vrr - slope * ver));
            rho = mr + slope * (e - me) - 4.5 * std::max(sd, 0.004);
            margin = 0.003;
        }
        return std::max(0.03, std::min(0.45, 1.0 - rho / double(P.match) + margin));
    };
    // Bands per couple.  A full batch (config 3: 811 couples) is bound by throughput: 11 bands of 1536 rows (66.8 ms; 7 of 2560:
    // 69.2; 17 of 1024: worse again).  A small one (a merge aligns a few hundred representatives) is bound by ONE couple's
    // critical path, (bands + strips - 1) tiles of rows / (4 bands) + 63 steps: more, shorter bands shorten it as long as the
    // waves the chip holds outnumber the tiles in flight.
    uint32_t few_limit = 128;  // couples in a launch up to which its tiles wait for rows instead of tiles (IOC_ALIGN_V2_FEW; 0: never)
    if (const char* e = getenv("IOC_ALIGN_V2_FEW")) few_limit = uint32_t(std::max(0, atoi(e)));
    uint32_t want_bands = 12;
    {
        int occ0 = 3;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, reinterpret_cast<const void*>(k_fwd2), 64 * V2_WAVES, 0) != hipSuccess || occ0 < 1) occ0 = 3;
        (void)hipGetLastError();
        const uint32_t waves = uint32_t(occ0) * uint32_t(n_cu) * uint32_t(V2_WAVES);
        want_bands = std::max(12u, std::min(uint32_t(V2_MAX_BANDS), (waves + ncouples - 1u) / std::max(1u, ncouples)));
        // (with tiles that follow their left neighbours row by row — v2_wait_rows — a strip starts ~120 steps after the one on its
        // left whatever the band's height: fewer, taller bands win again.  A second round of 31 couples of 16.7 kb: 33 bands of 512
        // rows 16.6 ms, 17 of 1024 rows 14.9, 11 of 1536 rows 16.2; without: 16.9 / 19.4 / 23.5)
        if (few_limit && ncouples <= few_limit) want_bands = std::min(want_bands, 17u);
    }
    if (const char* e = getenv("IOC_ALIGN_V2_BANDS")) want_bands = uint32_t(std::max(1, std::min(64, atoi(e))));
    std::vector<V2Couple> cps(ncouples);
    std::vector<V2PairCk> pck(np);
    std::vector<V2PairEnd> pend(np);
    std::vector<uint64_t> cwords(ncouples);  // arena words of a couple
    auto up4 = [](uint64_t x) { return (x + 3u) & ~uint64_t(3); };
    uint32_t lrow_total = 0, best_total = 0, max_strips = 1;
    uint64_t prof_total = 0;
    for (uint32_t k2 = 0; k2 < ncouples; ++k2) {
        V2Couple& cp = cps[k2];
        cp.pid[0] = order[2u * k2];
        cp.pid[1] = 2u * k2 + 1u < cnt ? order[2u * k2 + 1u] : 0xFFFFFFFFu;
        uint32_t nmax = 0, mmax = 0;
        for (int h = 0; h < 2; ++h)
            if (cp.pid[h] != 0xFFFFFFFFu) {
                nmax = std::max(nmax, dp[cp.pid[h]].n);
                mmax = std::max(mmax, dp[cp.pid[h]].m);
            }
        const uint32_t ncoarse = (nmax + CK2 - 1) / CK2;
        cp.nstrips = (mmax + 64u * FW_C - 1) / (64u * FW_C);
        // bands of whole coarse rows: enough of them for the anti-diagonals of the tile grid to keep the chip busy, each long
        // enough (>= 1024 rows = 256 steps) for the 63 steps of pipeline fill to stay small — except the first and the last
        // ones, which CAN be short (V2Couple::bstart; IOC_ALIGN_V2_RAMP=1): 1, 2, 4 coarse rows, then the regular height, then 4, 2, 1
        {
            // (a pair long enough for a corridor: shorter bands follow the diagonal more closely — config 3 with the corridor: 17 bands
            // of 1024 rows 45.9 ms, 11 of 1536 47.6, 22 of 1024 / 512 46.1)
            const uint32_t wb = (corridor_frac > 0.0 && ncoarse >= 16u && !getenv("IOC_ALIGN_V2_BANDS")) ? std::max(want_bands, 17u) : want_bands;
            const uint32_t reg = std::max<uint32_t>(want_bands > 16u ? 1u : 2u, (ncoarse + wb - 1) / wb);  // (coarse rows per band)
            std::vector<uint32_t> hts;
            uint32_t left = ncoarse;
            auto take = [&](uint32_t hgt) {
                hgt = std::min(hgt, left);
                if (hgt) hts.push_back(hgt);
                left -= hgt;
            };
            // (measured on config 3: the ramp costs more in pipeline fill than it wins — 69.9 against 69.2 ms at 7 regular bands:
            // off unless IOC_ALIGN_V2_RAMP=1)
            const bool ramp = ncoarse >= 4u * reg && getenv("IOC_ALIGN_V2_RAMP") != nullptr;
            std::vector<uint32_t> tail;
            if (ramp) {
                for (uint32_t hgt = 1; hgt < reg; hgt *= 2) take(hgt);
                for (uint32_t hgt = 1; hgt < reg && left > hgt; hgt *= 2) {
                    tail.push_back(hgt);
                    left -= hgt;
                }
            }
            while (left) take(reg);
            for (size_t x = tail.size(); x-- > 0;) hts.push_back(tail[x]);
            while (hts.size() > size_t(V2_MAX_BANDS)) {  // (very long pairs: merge from the middle)
                const size_t mid = hts.size() / 2;
                hts[mid - 1] += hts[mid];
                hts.erase(hts.begin() + long(mid));
            }
            cp.nbands = uint32_t(hts.size());
            cp.tpb = reg;
            uint32_t acc = 0;
            for (uint32_t b = 0; b <= uint32_t(V2_MAX_BANDS); ++b) {
                cp.bstart[b] = uint16_t(std::min<uint32_t>(acc, 0xFFFFu));
                if (b < cp.nbands) acc += hts[b];
            }
        }
        {   // the corridor (V2Couple): tiles that meet |row - column| <= B, if that leaves out a fifth of the grid or more
            cp.corridor = 0;
            for (uint32_t b = 0; b < uint32_t(V2_MAX_BANDS); ++b) {
                cp.plo[b] = 0;
                cp.phi[b] = uint16_t(std::min<uint32_t>(65535u, cp.nstrips - 1u));  // (V2Item::strip is 16 bits wide)
            }
            const uint32_t strip_cols = 64u * FW_C;
            double frac = 0.0;
            for (int h = 0; h < 2; ++h)
                if (cp.pid[h] != 0xFFFFFFFFu) frac = std::max(frac, pair_frac(dp[cp.pid[h]]));
            if (corridor_frac <= 0.0) frac = 0.0;
            const uint32_t B = uint32_t(frac * double(std::max(nmax, mmax)));
            cp.b0 = B;
            int64_t beff = INT64_MAX;
            uint32_t skipped = 0;
            uint16_t lo[V2_MAX_BANDS], hi[V2_MAX_BANDS];
            // (the bound behind the certificate: a column of an alignment scores `match` at most — no reward for mismatches or gaps)
            bool fits = corridor_frac > 0.0 && B >= strip_cols && cp.nstrips <= 65535u && cp.nbands <= uint32_t(V2_MAX_BANDS) && P.match > 0 && P.mismatch <= P.match &&
                        P.gap_extend >= 0;
            for (int h = 0; fits && h < 2; ++h)
                if (cp.pid[h] != 0xFFFFFFFFu) fits = dp[cp.pid[h]].gap_open >= 0;
            for (uint32_t b = 0; fits && b < cp.nbands; ++b) {
                const uint32_t r0 = uint32_t(cp.bstart[b]) * uint32_t(CK2), r1 = std::min(nmax, uint32_t(cp.bstart[b + 1u]) * uint32_t(CK2));
                const uint32_t c_lo = r0 > B ? r0 - B : 0u, c_hi = std::min<uint64_t>(mmax, uint64_t(r1) + B);
                if (c_hi <= c_lo) {  // (a band wholly below the last column's corridor: rows the longer query has alone)
                    fits = false;
                    break;
                }
                lo[b] = uint16_t(c_lo / strip_cols);
                hi[b] = uint16_t((c_hi - 1u) / strip_cols);
                skipped += lo[b] + (cp.nstrips - 1u - hi[b]);
                if (lo[b] > 0) beff = std::min<int64_t>(beff, int64_t(r0) - int64_t(lo[b]) * strip_cols);
                if ((uint64_t(hi[b]) + 1u) * strip_cols < mmax) beff = std::min<int64_t>(beff, int64_t(hi[b] + 1u) * strip_cols - int64_t(r1));
            }
            // the probe (V2Couple): the first band that ends at row 2048 or below, the strip its last diagonal cell is in; every tile
            // above and to the left of it must be inside the corridor with nothing missing around it, both pairs must reach it
            uint32_t pb = 0, ps = 0;
            if (fits) {
                while (pb + 1u < cp.nbands && uint32_t(cp.bstart[pb + 1u]) * uint32_t(CK2) < probe_rows) ++pb;  // (3072 rows: a probe launch of 4.2 ms; 2048: 2.3)
                const uint32_t R = uint32_t(cp.bstart[pb + 1u]) * uint32_t(CK2);
                ps = (R - 1u) / strip_cols;
                fits = pb + 1u < cp.nbands && R >= probe_rows;
                for (uint32_t b = 0; fits && b <= pb; ++b) fits = lo[b] == 0 && hi[b] >= ps;
                for (int h = 0; fits && h < 2; ++h)
                    if (cp.pid[h] != 0xFFFFFFFFu) fits = dp[cp.pid[h]].n >= R && dp[cp.pid[h]].m > ps * strip_cols;
            }
            cp.probe_band = uint16_t(pb);
            cp.probe_strip = uint16_t(ps);
            if (fits && skipped * 5u >= cp.nbands * cp.nstrips && beff > 0 && beff < INT64_MAX) {
                cp.corridor = 1;
                for (uint32_t b = 0; b < cp.nbands; ++b) {
                    cp.plo[b] = lo[b];
                    cp.phi[b] = hi[b];
                }
                tiles_skipped += skipped;
            }
            tiles_all += uint64_t(cp.nbands) * cp.nstrips;
            for (int h = 0; h < 2; ++h)
                if (cp.pid[h] != 0xFFFFFFFFu) {
                    const AlnPairDev& d = dp[cp.pid[h]];
                    // (a path through a skipped cell has at most max(n, m) - Beff - 1 diagonal steps, each worth `match` at most)
                    pend_cert[cp.pid[h]] = cp.corridor ? int32_t(std::min<int64_t>(INT32_MAX, int64_t(P.match) * std::max<int64_t>(0, int64_t(std::max(d.n, d.m)) - beff - 1))) : INT32_MIN;
                }
        }
        cp.mpad = uint32_t((uint64_t(mmax) + 15u) & ~uint64_t(15));
        cp.nq = (nmax + 3u) / 4u;
        const uint64_t ncr = (nmax - 1u) / CK2, ncc = (mmax - 1u) / CK2;
        uint64_t w = 0;
        cp.rdat = w;
        w += up4(ncr * 2u * cp.mpad);
        cp.rbase = w;
        w += up4(ncr * (cp.mpad / 16u) * 2u);
        cp.cdat = w;
        w += up4(ncc * uint64_t(cp.nq) * 8u);
        cp.cbase = w;
        w += up4(ncc * uint64_t(cp.nq) * 2u);
        cwords[k2] = w;
        cp.prof = 0;  // (uint4 entries; placed per slice below: the profiles of a slice's couples count against the budget with its checkpoints)
        max_strips = std::max(max_strips, cp.nstrips);
        for (int h = 0; h < 2; ++h) {
            cp.lrow0[h] = lrow_total;
            cp.best0[h] = best_total;
            if (cp.pid[h] == 0xFFFFFFFFu) continue;
            const AlnPairDev& d = dp[cp.pid[h]];
            const uint32_t nstr = (d.m + 64u * FW_C - 1) / (64u * FW_C);
            pend[cp.pid[h]] = V2PairEnd{lrow_total, best_total, cp.nbands, k2, pend_cert[cp.pid[h]]};
            lrow_total += nstr * 64u;
            best_total += cp.nbands;
        }
    }
    // slices of couples whose checkpoints fit the budget together
    std::vector<std::pair<uint32_t, uint32_t>> slices;
    uint64_t arena_words = 0;
    for (uint32_t first = 0; first < ncouples;) {
        uint64_t used = 0;
        uint32_t k2 = first;
        uint64_t prof_used = 0;  // uint4 entries of query profiles in this slice (they used to be made for every couple of the call at once)
        while (k2 < ncouples) {
            const uint64_t prof_k = uint64_t(cps[k2].nstrips) * 2u * 5u * 64u;
            if (k2 != first && (used + cwords[k2]) * 4ull + (prof_used + prof_k) * sizeof(uint4) > budget) break;
            V2Couple& cp = cps[k2];
            cp.prof = prof_used;
            prof_used += prof_k;
            cp.rdat += used;
            cp.rbase += used;
            cp.cdat += used;
            cp.cbase += used;
            for (int h = 0; h < 2; ++h)
                if (cp.pid[h] != 0xFFFFFFFFu) pck[cp.pid[h]] = V2PairCk{cp.rdat, cp.rbase, cp.cdat, cp.cbase, cp.mpad, cp.nq, uint32_t(h), k2};
            used += cwords[k2];
            ++k2;
        }
        arena_words = std::max(arena_words, used);
        prof_total = std::max(prof_total, prof_used);
        slices.emplace_back(first, k2 - first);
        first = k2;
    }
    const auto t_res0 = std::chrono::steady_clock::now();
    if ((r = reserve(c, c->a_ck, size_t(arena_words) * 4)) != IOC_OK) return r;
    c->tm.align_arena_bytes = int64_t(arena_words) * 4;
    c->tm.align_slices = int32_t(slices.size());
    c->tm.align_version = 2;
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc]   aligner v2: %u couples, checkpoint arena %.1f MB (%zu slice(s)) reserved in %.3f ms; corridor: %llu of %llu tiles skipped\n", ncouples,
                double(arena_words) * 4e-6, slices.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_res0).count(),
                (unsigned long long)tiles_skipped, (unsigned long long)tiles_all);
    // flags and items per slice; the static tables once
    uint32_t max_items = 0, max_flags = 0, max_pairs = 0;
    std::vector<std::vector<V2Item>> items(slices.size());
    std::vector<uint32_t> n_probe(slices.size(), 0);  // (the probe launch's tiles come first in a slice's list)
    for (size_t si = 0; si < slices.size(); ++si) {
        uint32_t nf = 0;
        auto& it = items[si];
        uint32_t maxdiag = 0;
        size_t total_items = 0;
        for (uint32_t k2 = slices[si].first; k2 < slices[si].first + slices[si].second; ++k2) {
            V2Couple& cp = cps[k2];
            cp.flag0 = nf;
            nf += cp.nbands * cp.nstrips;
            total_items += size_t(cp.nbands) * cp.nstrips;
            maxdiag = std::max(maxdiag, cp.nbands + cp.nstrips - 2u);
        }
        // anti-diagonal by anti-diagonal, couples side by side: every tile comes after the tile above it and the tile to its left
        // (generated in that order: sorting 150 000 items cost 4 ms of host time per call)
        it.reserve(total_items + 64u * slices[si].second);
        // first the probe's tiles (couples with a corridor: the top left corner of the grid, V2Couple), in the same order
        for (uint32_t dg = 0; dg <= maxdiag; ++dg) {
            bool any = false;
            for (uint32_t k2 = slices[si].first; k2 < slices[si].first + slices[si].second; ++k2) {
                const V2Couple& cp = cps[k2];
                if (!cp.corridor || dg > uint32_t(cp.probe_band) + cp.probe_strip) continue;
                any = true;
                const uint32_t b_lo = dg >= uint32_t(cp.probe_strip) + 1u ? dg - cp.probe_strip : 0u, b_hi = std::min<uint32_t>(dg, cp.probe_band);
                for (uint32_t b = b_lo; b <= b_hi; ++b) it.push_back(V2Item{k2, uint16_t(b), uint16_t(dg - b)});
            }
            if (!any && dg > 64u) break;
        }
        n_probe[si] = uint32_t(it.size());
        for (uint32_t dg = 0; dg <= maxdiag; ++dg)
            for (uint32_t k2 = slices[si].first; k2 < slices[si].first + slices[si].second; ++k2) {
                const V2Couple& cp = cps[k2];
                if (dg > cp.nbands + cp.nstrips - 2u) continue;
                const uint32_t b_lo = dg >= cp.nstrips ? dg - cp.nstrips + 1u : 0u, b_hi = std::min(dg, cp.nbands - 1u);
                for (uint32_t b = b_lo; b <= b_hi; ++b) it.push_back(V2Item{k2, uint16_t(b), uint16_t(dg - b)});
            }
        max_items = std::max<uint32_t>(max_items, uint32_t(it.size()));
        max_flags = std::max(max_flags, nf);
        max_pairs = std::max(max_pairs, std::min(cnt, 2u * (slices[si].first + slices[si].second)) - 2u * slices[si].first);
    }
    const size_t tab_bytes = size_t(ncouples) * sizeof(V2Couple) + size_t(np) * (sizeof(V2PairCk) + sizeof(V2PairEnd)) + size_t(max_items) * sizeof(V2Item) + 256;
    if ((r = reserve(c, c->a_cko, tab_bytes)) != IOC_OK) return r;
    uint8_t* tb = static_cast<uint8_t*>(c->a_cko.p);
    V2Couple* d_cps = reinterpret_cast<V2Couple*>(tb);
    V2PairCk* d_pck = reinterpret_cast<V2PairCk*>(tb + size_t(ncouples) * sizeof(V2Couple));
    V2PairEnd* d_pend = reinterpret_cast<V2PairEnd*>(reinterpret_cast<uint8_t*>(d_pck) + size_t(np) * sizeof(V2PairCk));
    V2Item* d_items = reinterpret_cast<V2Item*>(reinterpret_cast<uint8_t*>(d_pend) + size_t(np) * sizeof(V2PairEnd));
    ACHK(c, hipMemcpyAsync(d_cps, cps.data(), size_t(ncouples) * sizeof(V2Couple), hipMemcpyHostToDevice, s));
    ACHK(c, hipMemcpyAsync(d_pck, pck.data(), size_t(np) * sizeof(V2PairCk), hipMemcpyHostToDevice, s));
    ACHK(c, hipMemcpyAsync(d_pend, pend.data(), size_t(np) * sizeof(V2PairEnd), hipMemcpyHostToDevice, s));
    // the query profiles of every (couple, strip), once per batch
    if ((r = reserve(c, c->a_prof, size_t(prof_total) * sizeof(uint4) + 256)) != IOC_OK) return r;
    // (the profiles themselves are made slice by slice, in front of the slice's forward pass)
    if ((r = reserve(c, c->a_ends2, size_t(np) * sizeof(int4))) != IOC_OK) return r;
    if ((r = reserve(c, c->a_lrow, (size_t(lrow_total) + best_total + 16) * sizeof(int2))) != IOC_OK) return r;
    int2* d_lrow = static_cast<int2*>(c->a_lrow.p);
    int2* d_best = d_lrow + lrow_total;
    // [queue][err][pad ...][flags][rows of a tile's right edge that are out (v2_wait_rows)][ovf per pair][computed cells, 8-byte aligned]
    const size_t ctl_words = ((16 + 2 * size_t(max_flags) + np + 1) & ~size_t(1)) + 2;
    if ((r = reserve(c, c->a_xflags, ctl_words * 4)) != IOC_OK) return r;
    uint32_t* d_ctl = static_cast<uint32_t*>(c->a_xflags.p);
    // the traceback in two launches: walks that need more than `deadline` blocks go on in the second one, with helper waves
    // (IOC_TRACE2_DEADLINE=0: one launch).  Per pair: its own scratch and 16 words of parked state and records.
    uint32_t deadline = 4;  // (blocks of slack on top of what a pair of one transcript needs to be decided: 2 per 512 windows of its threshold)
    if (const char* e = getenv("IOC_TRACE2_DEADLINE")) deadline = uint32_t(std::max(0, atoi(e)));
    // ... or that are still undecided after this many cycles of the first launch (s_memtime; 0: the block count alone)
    unsigned long long deadline_cycles = 0;
    if (const char* e = getenv("IOC_TRACE2_CYCLES")) deadline_cycles = (unsigned long long)std::max(0.0, atof(e));
    // (the helper buffers belong to the second launch's WORKGROUPS — one per compute unit — not to the pairs: 0.66 MB each)
    const uint32_t help_wgs = deadline ? std::min<uint32_t>(max_pairs, uint32_t(n_cu)) : 0u;
    const size_t n_scratch = size_t(max_pairs) + size_t(help_wgs) * V2_NHELP * V2_HBUF;
    const size_t resume_words = size_t(np) * V2_RESUME_WORDS + 2u * (size_t(max_pairs) + 1u) + 4u;  // (parked state, the two lists of walks for the helper launches)
    if ((r = reserve(c, c->a_bnd, n_scratch * sizeof(V2Scratch) + resume_words * 4)) != IOC_OK) return r;
    V2Scratch* d_scratch = static_cast<V2Scratch*>(c->a_bnd.p);
    V2Scratch* d_hscratch = d_scratch + max_pairs;
    uint32_t* d_resume = reinterpret_cast<uint32_t*>(d_scratch + n_scratch);
    uint32_t* d_park = d_resume + size_t(np) * V2_RESUME_WORDS;  // [count][pair slots of the slice]: parked by the first traceback launch
    uint32_t* d_early = d_park + size_t(max_pairs) + 1u;         // ... and sent to the helper launch by k_fwd2_ends (wrong candidates, by their score)
    uint32_t* d_gate = d_early + size_t(max_pairs) + 1u;         // workgroups of that launch that have started
    // The helper launch for those runs on a side stream BESIDE the first traceback launch (IOC_TRACE2_EARLY=0: everything goes
    // through the first launch, as before): the walks of the wrong candidates are the longest of the batch, and the first launch —
    // one pair's latency long, with the chip half empty — no longer waits for them to reach their deadline first.
    const bool route = deadline && !(getenv("IOC_TRACE2_EARLY") && atoi(getenv("IOC_TRACE2_EARLY")) == 0);
    if (route && !c->side_stream) {
        ACHK(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        for (auto& e : c->ev_side) ACHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(k_fwd2), 64 * V2_WAVES, 0) != hipSuccess) occ = 0;
    (void)hipGetLastError();
    if (occ < 1) occ = 3;
    struct EventSet {  // (destroyed on every way out, the ACHK returns included)
        std::vector<hipEvent_t> v;
        ~EventSet()
        {
            for (auto& e : v)
                if (e) (void)hipEventDestroy(e);
        }
    } evset;
    evset.v.assign(slices.size() * 3, nullptr);
    std::vector<hipEvent_t>& evs = evset.v;
    for (auto& e : evs) ACHK(c, hipEventCreate(&e));
    size_t evi = 0;
    bool bad = false;
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc]   aligner v2: planning (couples, corridors, tile lists, tables) took the host %.3f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count());
    for (size_t si = 0; si < slices.size() && !bad; ++si) {
        const uint32_t first_pair = 2u * slices[si].first, n_pairs = std::min(cnt, 2u * (slices[si].first + slices[si].second)) - first_pair;
        const uint32_t n_items = uint32_t(items[si].size());
        // (few couples: bound by ONE grid's critical path — tiles follow their left neighbours row by row, v2_wait_rows)
        const bool few_couples = slices[si].second <= few_limit;
        // (experiment: every launch lets a tile follow its left neighbour, which publishes its progress every IOC_ALIGN_V2_PROG_ALL blocks of 64 rows)
        const uint32_t prog_all = getenv("IOC_ALIGN_V2_PROG_ALL") ? uint32_t(std::max(1, atoi(getenv("IOC_ALIGN_V2_PROG_ALL")))) : 0u;
        const bool use_prog = few_couples || prog_all != 0;
        const uint32_t prog_every = few_couples ? 1u : prog_all;
        ACHK(c, hipMemcpyAsync(d_items, items[si].data(), size_t(n_items) * sizeof(V2Item), hipMemcpyHostToDevice, s));
        ACHK(c, hipMemcpyAsync(d_cps + slices[si].first, cps.data() + slices[si].first, size_t(slices[si].second) * sizeof(V2Couple), hipMemcpyHostToDevice, s));  // (flag0)
        hipLaunchKernelGGL(k_fwd2_prof, dim3(slices[si].second, max_strips), dim3(128), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p), d_cps + slices[si].first,
                           static_cast<const uint8_t*>(c->a_pool.p), P, static_cast<uint4*>(c->a_prof.p));
        ACHK(c, hipGetLastError());
        ACHK(c, hipMemsetAsync(d_ctl, 0, ctl_words * 4, s));
        ACHK(c, hipMemsetAsync(d_resume, 0, resume_words * 4, s));  // (parked state and the two lists of walks; k_fwd2_ends writes into them)
        if (tiles_skipped)  // (the bests of the last row and the last column that a skipped tile does not write: far below any score)
            ACHK(c, hipMemsetAsync(d_lrow, 0x80, (size_t(lrow_total) + best_total) * sizeof(int2), s));
        if (getenv("IOC_ALIGN_V2_FAKE_TIMEOUT")) {  // (tests: as if a bounded wait had run out — every later wait gives up at once,
            const uint32_t one = 1;                 // tiles run on whatever is there, the host must fall back to version 1)
            ACHK(c, hipMemcpyAsync(d_ctl + 1, &one, 4, hipMemcpyHostToDevice, s));
        }
        ACHK(c, hipEventRecord(evs[evi++], s));
        // persistent waves: as many workgroups as the chip holds, but no more waves than tiles
        if (n_probe[si]) {  // the probe launch and its verdicts (V2Couple): all on the device, nothing comes back to the host
            const uint32_t n_wg0 = std::max(1u, std::min(uint32_t(occ) * uint32_t(n_cu), (n_probe[si] + V2_WAVES - 1) / V2_WAVES));
            hipLaunchKernelGGL(k_fwd2, dim3(n_wg0), dim3(64 * V2_WAVES), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p), d_cps, d_items, n_probe[si],
                               d_ctl, d_ctl + 16, d_ctl + 1, static_cast<const uint8_t*>(c->a_pool.p), P, static_cast<uint32_t*>(c->a_ck.p), d_lrow,
                               d_best, d_ctl + 16 + 2 * max_flags, guard, static_cast<const uint4*>(c->a_prof.p), reinterpret_cast<unsigned long long*>(d_ctl + ctl_words - 2), use_prog ? d_ctl + 16 + max_flags : static_cast<uint32_t*>(nullptr), prog_every);
            ACHK(c, hipGetLastError());
            hipLaunchKernelGGL(k_fwd2_probe, dim3(slices[si].second), dim3(64), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p), d_cps + slices[si].first, d_pend,
                               static_cast<const uint32_t*>(c->a_ck.p), P);
            ACHK(c, hipGetLastError());
            ACHK(c, hipMemsetAsync(d_ctl, 0, 4, s));  // (the queue's counter; the flags of the probe's tiles stay)
        }
        const uint32_t n_main = n_items - n_probe[si];
        // Workgroups per compute unit.  A tile waits for the tile above and the tile to its left, so a couple's grid offers only
        // its current anti-diagonal — with the corridor 2 - 3 tiles of a 16.7 kb pair — and a tile takes 0.5 - 0.9 ms: the pass is
        // bound by tiles in flight as much as by throughput (config 3: 37 ms for a critical path of ~18).  Waves beyond the tiles
        // that can run sit on a tile whose neighbours are still being computed, and they sit unevenly: a SIMD with three busy
        // waves next to one with a single busy wave and two waiting.  So: as many workgroups per CU as the list's tiles per
        // anti-diagonal keep busy, the same number on every CU (config 3 with the corridor: 2 per CU 37.2 ms, 3 per CU 39.8,
        // 2.25 per CU 40.6; every tile: 3 per CU 67.6, 2 per CU 72.7; a quarter of the batch: 1 per CU 22.7, 3 per CU 25.1).
        uint32_t per_cu = uint32_t(occ);
        {
            uint64_t tiles = 0;
            uint32_t diags = 1;
            for (uint32_t k2 = slices[si].first; k2 < slices[si].first + slices[si].second; ++k2) {
                const V2Couple& cp = cps[k2];
                for (uint32_t b = 0; b < cp.nbands; ++b) tiles += uint32_t(cp.phi[b]) - uint32_t(cp.plo[b]) + 1u;
                diags = std::max(diags, cp.nbands + cp.nstrips - 1u);
            }
            const double x = double(tiles) / double(diags) / double(uint32_t(n_cu) * uint32_t(V2_WAVES));
            if (!few_couples) per_cu = x < 1.0 ? 1u : x < 2.75 ? std::min(2u, uint32_t(occ)) : uint32_t(occ);
        }
        uint32_t n_wg = std::max(1u, std::min(per_cu * uint32_t(n_cu), (n_main + V2_WAVES - 1) / V2_WAVES));
        if (const char* e = getenv("IOC_ALIGN_V2_WGS")) n_wg = std::max(1u, std::min(uint32_t(occ) * uint32_t(n_cu), uint32_t(atoi(e))));  // (experiments)
        // (two workgroups per CU: the build of the kernel that may use the registers of the third)
        auto* fwd_kernel = (per_cu <= 2 && !getenv("IOC_ALIGN_V2_NO_W2")) ? k_fwd2_w2 : k_fwd2;
        hipLaunchKernelGGL(fwd_kernel, dim3(n_wg), dim3(64 * V2_WAVES), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p), d_cps, d_items + n_probe[si], n_main,
                               d_ctl, d_ctl + 16, d_ctl + 1, static_cast<const uint8_t*>(c->a_pool.p), P, static_cast<uint32_t*>(c->a_ck.p), d_lrow,
                               d_best, d_ctl + 16 + 2 * max_flags, guard, static_cast<const uint4*>(c->a_prof.p), reinterpret_cast<unsigned long long*>(d_ctl + ctl_words - 2), use_prog ? d_ctl + 16 + max_flags : static_cast<uint32_t*>(nullptr), prog_every);
        ACHK(c, hipGetLastError());
        hipLaunchKernelGGL(k_fwd2_ends, dim3(n_pairs), dim3(64), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p), d_cps, d_order + first_pair, n_pairs,
                           d_pend, d_lrow, d_best, d_ctl + 16 + 2 * max_flags, static_cast<int4*>(c->a_ends2.p), d_resume, d_early, route ? int(P.match) : 0,
                           std::min<uint32_t>(n_pairs, help_wgs));
        ACHK(c, hipGetLastError());
        ACHK(c, hipEventRecord(evs[evi++], s));
        if (route) {  // (issued before the first launch: its workgroups — a walker and ten helpers each — take their places first)
            ACHK(c, hipEventRecord(c->ev_side[0], s));
            ACHK(c, hipStreamWaitEvent(c->side_stream, c->ev_side[0], 0));
            hipLaunchKernelGGL(k_trace2_help, dim3(std::min<uint32_t>(n_pairs, help_wgs)), dim3(64 * V2_HWAVES), 0, c->side_stream,
                               static_cast<const AlnPairDev*>(c->a_pairs.p), d_order + first_pair, static_cast<const uint8_t*>(c->a_pool.p), P,
                               static_cast<const uint32_t*>(c->a_ck.p), d_cps, d_pck, static_cast<const int4*>(c->a_ends2.p), d_scratch, d_hscratch, d_resume,
                               d_early, d_score, d_count, n_pairs, 2u, d_gate);
            ACHK(c, hipGetLastError());
            ACHK(c, hipEventRecord(c->ev_side[1], c->side_stream));
            hipLaunchKernelGGL(k_trace2_gate, dim3(1), dim3(1), 0, s, d_early, d_gate, std::min<uint32_t>(n_pairs, help_wgs));
            ACHK(c, hipGetLastError());
        }
        hipLaunchKernelGGL(k_trace2, dim3((n_pairs + TR_WAVES - 1) / TR_WAVES), dim3(64 * TR_WAVES), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p),
                           d_order + first_pair, static_cast<const uint8_t*>(c->a_pool.p), P, static_cast<const uint32_t*>(c->a_ck.p), d_cps, d_pck,
                           static_cast<const int4*>(c->a_ends2.p), d_scratch, d_resume, d_park, d_score, d_count, n_pairs, deadline, deadline_cycles);
        ACHK(c, hipGetLastError());
        if (route) ACHK(c, hipStreamWaitEvent(s, c->ev_side[1], 0));  // (the two helper launches share the helpers' buffers)
        if (deadline) {  // (one workgroup per compute unit: a walker and its helpers fill one; more parked walks than that take turns)
            hipLaunchKernelGGL(k_trace2_help, dim3(std::min<uint32_t>(n_pairs, help_wgs)), dim3(64 * V2_HWAVES), 0, s,
                               static_cast<const AlnPairDev*>(c->a_pairs.p), d_order + first_pair, static_cast<const uint8_t*>(c->a_pool.p), P,
                               static_cast<const uint32_t*>(c->a_ck.p), d_cps, d_pck, static_cast<const int4*>(c->a_ends2.p), d_scratch, d_hscratch, d_resume,
                               d_park, d_score, d_count, n_pairs, 1u, nullptr);
            ACHK(c, hipGetLastError());
        }
        ACHK(c, hipEventRecord(evs[evi++], s));
        if (getenv("IOC_V2_PROGRESS")) {  // debug watchdog: where are the kernels after 10 s?  (build with -DIOC_V2_MARKS)
            const auto tw = std::chrono::steady_clock::now();
            while (hipStreamQuery(s) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count() > 10.0) {
                    hipStream_t s2 = nullptr;
                    uint32_t pg[16] = {0};
                    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
                    (void)hipMemcpyAsync(pg, d_ctl, sizeof pg, hipMemcpyDeviceToHost, s2);
                    (void)hipStreamSynchronize(s2);
                    fprintf(stderr, "[ioc] v2 WATCHDOG: queue %u err %u | dequeued %u, past waits %x, nsteps %u, step %u, after loop %u, published %u, wave exits %x %x %x %x, end marks %u %u %u\n",
                            pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], pg[6], pg[7], pg[8], pg[9], pg[10], pg[11], pg[12], pg[13], pg[14]);
                    fflush(stderr);
                    abort();
                }
            }
        }
        uint32_t xe = 0;
        unsigned long long cells_done = 0;
        ACHK(c, hipMemcpyAsync(&xe, d_ctl + 1, 4, hipMemcpyDeviceToHost, s));
        ACHK(c, hipMemcpyAsync(&cells_done, d_ctl + ctl_words - 2, 8, hipMemcpyDeviceToHost, s));
        ACHK(c, hipStreamSynchronize(s));
        c->tm.n_align_cells_computed += int64_t(cells_done);
        if (getenv("IOC_V2_PROGRESS")) {  // (marks build: cycles the waves spent waiting for tiles / alive / the longest-lived wave)
            unsigned long long t[3] = {0, 0, 0};
            uint32_t pg[16] = {0};
            ACHK(c, hipMemcpy(pg, d_ctl, sizeof pg, hipMemcpyDeviceToHost));
            memcpy(&t[0], &pg[12], 8);
            memcpy(&t[1], &pg[14], 8);
            memcpy(&t[2], &pg[10], 8);
            fprintf(stderr, "[ioc] v2 marks: waiting %.3e cycles of %.3e wave-cycles alive (%.1f %%), longest wave %.3e cycles, %u workgroups\n", double(t[0]), double(t[1]),
                    100.0 * double(t[0]) / std::max(1.0, double(t[1])), double(t[2]), n_wg);
        }
        if (getenv("IOC_V2_TRACE_RATES")) {  // developer aid: score per base against the summed error rate, pair by pair (the corridor's calibration)
            std::vector<int4> he(np);
            ACHK(c, hipMemcpy(he.data(), c->a_ends2.p, size_t(np) * sizeof(int4), hipMemcpyDeviceToHost));
            for (uint32_t x = 0; x < n_pairs; ++x) {
                const uint32_t pid = order[first_pair + x];
                const AlnPairDev& d = dp[pid];
                fprintf(stderr, "[rate] e %.5f n %u m %u score %d per_base %.5f go %d flags %d corridor %u\n", double(d.e_sum), d.n, d.m, he[pid].x,
                        double(he[pid].x) / double(std::max(d.n, d.m)), d.gap_open, he[pid].w, cps[pend[pid].couple].corridor);
            }
        }
        if (getenv("IOC_V2_TRACE_TIMES")) {  // developer aid: the slowest walks of the slice (k_trace2 ends when its slowest wave does)
            std::vector<uint32_t> rw(size_t(np) * V2_RESUME_WORDS);
            ACHK(c, hipMemcpy(rw.data(), d_resume, rw.size() * 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> ids(n_pairs);
            for (uint32_t x = 0; x < n_pairs; ++x) ids[x] = order[first_pair + x];
            // (first launch + second launch: the second starts when the first has ended, so the sum orders the pairs by what they cost)
            auto rec = [&](uint32_t pid, uint32_t w) { return rw[size_t(pid) * V2_RESUME_WORDS + w]; };
            auto cost = [&](uint32_t pid) { return (double(rec(pid, 8)) + double(rec(pid, 12))) * 256.0; };
            std::sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
            double sum = 0;
            uint32_t parked = 0;
            for (uint32_t pid : ids) {
                sum += cost(pid);
                parked += rec(pid, 6) ? 1u : 0u;
            }
            fprintf(stderr, "[ioc] k_trace2: %u pairs, %u walks went on in the second launch (deadline %u blocks); mean %.3e cycles (s_memtime), slowest first:\n", n_pairs, parked,
                    deadline, sum / n_pairs);
            auto line = [&](const char* tag, uint32_t pid) {
                const AlnPairDev& d = dp[pid];
                fprintf(stderr, "[ioc]   %s pair %u: %.3e + %.3e cycles, n %u m %u, blocks %u (%u on the diagonal) + %u (%u by helpers), tiles %u + %u, gap steps %u, windows %u\n", tag, pid,
                        double(rec(pid, 8)) * 256.0, double(rec(pid, 12)) * 256.0, d.n, d.m, rec(pid, 9) & 0xFFFFu, rec(pid, 9) >> 16, rec(pid, 13) & 0xFFFFu, rec(pid, 13) >> 16,
                        rec(pid, 10), rec(pid, 14), rec(pid, 11), rec(pid, 15));
            };
            {   // how well does the forward score tell the walks that get parked (wrong candidates) from the others?  score / shorter length
                std::vector<int4> he(np);
                ACHK(c, hipMemcpy(he.data(), c->a_ends2.p, size_t(np) * sizeof(int4), hipMemcpyDeviceToHost));
                double pk_max = -1e9, un_min = 1e9, pk_sum = 0, un_sum = 0;
                uint32_t npk = 0, nun = 0;
                std::vector<double> pks, uns;
                for (uint32_t pid : ids) {
                    const AlnPairDev& d = dp[pid];
                    const double q = double(he[pid].x) / double(std::max(1u, std::min(d.n, d.m)));
                    if (rec(pid, 6)) {
                        pk_max = std::max(pk_max, q), pk_sum += q, ++npk;
                        pks.push_back(q);
                    } else {
                        un_min = std::min(un_min, q), un_sum += q, ++nun;
                        uns.push_back(q);
                    }
                }
                std::sort(pks.begin(), pks.end());
                std::sort(uns.begin(), uns.end());
                auto qt = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, size_t(f * double(v.size())))]; };
                fprintf(stderr, "[ioc]   forward score / shorter length: parked walks (%u) mean %.3f, 50 %% %.3f, 90 %% %.3f, 99 %% %.3f, max %.3f | others (%u) min %.3f, 1 %% %.3f, 10 %% %.3f, mean %.3f\n", npk,
                        npk ? pk_sum / npk : 0.0, qt(pks, 0.5), qt(pks, 0.9), qt(pks, 0.99), pk_max, nun, un_min, qt(uns, 0.01), qt(uns, 0.1), nun ? un_sum / nun : 0.0);
            }
            for (uint32_t x = 0; x < std::min(n_pairs, 10u); ++x) line("", ids[x]);
            for (uint32_t qx : {n_pairs / 4u, n_pairs / 2u, 3u * n_pairs / 4u}) line("rank", ids[qx]);
        }
        if (xe) bad = true;
    }
    for (size_t x = 0; x + 2 < evi + 0 && x + 2 < evs.size(); x += 3) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, evs[x], evs[x + 1]) == hipSuccess) c->tm.ms_align_fwd += a;
        if (hipEventElapsedTime(&b, evs[x + 1], evs[x + 2]) == hipSuccess) c->tm.ms_align_trace += b;
    }
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc]   aligner v2: forward %.3f ms, traceback %.3f ms (device, all slices so far)%s\n", c->tm.ms_align_fwd,
                c->tm.ms_align_trace, bad ? " — a wait ran out: falling back to version 1" : "");
    *fell_back = bad;
    return IOC_OK;
}

}  // namespace

#define ACHK_UNUSED(c, call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// set while pairs the packed kernel gave up on are re-run
static thread_local bool g_force32 = false;

extern "C" {

int ioc_align_set_pool(ioc_ctx* c, int32_t n_seqs, const char* seqs, const int64_t* offs)
{
    if (!c || n_seqs < 0 || (n_seqs > 0 && (!seqs || !offs))) return IOC_ERR_ARG;
    ACHK(c, hipSetDevice(c->device));
    c->res_pool_ready = false;
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
    // which sequences hold something else than A C G T
    if ((r = reserve(c, c->a_ends, size_t(n_seqs + 1) * 8 + size_t(n_seqs))) != IOC_OK) return r;
    int64_t* d_offs = static_cast<int64_t*>(c->a_ends.p);
    uint8_t* d_flags = reinterpret_cast<uint8_t*>(d_offs + n_seqs + 1);
    ACHK(c, hipMemcpyAsync(d_offs, offs, size_t(n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_seq_flags, dim3(uint32_t(n_seqs)), dim3(256), 0, c->stream,
                       static_cast<const uint8_t*>(c->a_pool.p), d_offs, d_flags);
    ACHK(c, hipGetLastError());
    c->aln_other.assign(size_t(n_seqs), 0);
    ACHK(c, hipMemcpyAsync(c->aln_other.data(), d_flags, size_t(n_seqs), hipMemcpyDeviceToHost, c->stream));
    ACHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_align_set_verdict_threshold(ioc_ctx* c, double aligned_threshold)
{
    if (!c) return IOC_ERR_ARG;
    c->aln_verdict_thr = aligned_threshold > 0.0 ? aligned_threshold : -1.0;
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
        d.e_sum = float(a.e);
        d.ilimit = il;
        d.rc = a.ref_revcomp ? 1u : 0u;
        d.stop_at = 0;
        if (c->aln_verdict_thr > 0.0 && n > 0) {
            // smallest count a with double(a) / double(n) >= threshold: the comparison the caller will make (getAlnRatio / slen)
            double a0 = std::ceil(c->aln_verdict_thr * double(n));
            while (a0 > 0 && (a0 - 1.0) / double(n) >= c->aln_verdict_thr) a0 -= 1.0;
            while (a0 / double(n) < c->aln_verdict_thr) a0 += 1.0;
            d.stop_at = a0 >= 1.0 && a0 < 4.0e9 ? uint32_t(a0) : 0u;
        }
        {   // query-profile kernel: four-letter sequences, diagonal increments that fit a byte
            const int gd = d.gap_open - gap_extend, cm = match + 2 * gap_extend + gd, cx = mismatch + 2 * gap_extend + gd;
            const bool ok = !c->aln_other[size_t(a.query)] && !c->aln_other[size_t(a.ref)] && cm >= 0 && cm <= 255 &&
                            cx >= 0 && cx <= 255 && !getenv("IOC_ALIGN_NO_PROFILE");
            d.pad = ok ? 1u : 0u;
        }
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
    // Pairs with the same tile grid (coarse rows x strips) in the order of their summed error rates: a couple's corridor is as wide as
    // its less similar pair needs (align_v2_run), so like goes with like.  Order only.
    if (!getenv("IOC_ALIGN_NO_ESORT")) {
        auto rows = [&](uint32_t x) { return uint64_t((dp[x].n + 511u) / 512u); };
        auto cols = [&](uint32_t x) { return uint64_t((dp[x].m + 511u) / 512u); };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint64_t tx = rows(x) * cols(x), ty = rows(y) * cols(y);
            if (tx != ty) return tx > ty;  // (heaviest first, as before — by tiles)
            if (rows(x) != rows(y)) return rows(x) > rows(y);
            return dp[x].e_sum > dp[y].e_sum;
        });
    }
    // The caller's similarity hints (ioc_aln_pair::reserved, 0 = none): pairs hinted far below the batch's median — a quarter of
    // it — are taken for unrelated and sorted behind the others, so that version 2 couples them with each other: a couple with
    // an unrelated pair gets no corridor, and a pair of one transcript coupled with it pays for every tile too (config 3: 19 %
    // of the couples hold a wrong candidate when they are coupled by size alone, 10 % of the pairs are one).  Order only.
    {
        std::vector<int32_t> hp;
        for (uint32_t x = 0; x < np; ++x)
            if (pairs[back[x]].reserved > 0) hp.push_back(pairs[back[x]].reserved);
        if (hp.size() >= 8 && !getenv("IOC_ALIGN_NO_HINTS")) {
            std::nth_element(hp.begin(), hp.begin() + long(hp.size() / 2), hp.end());
            const int64_t med = hp[hp.size() / 2];
            std::stable_partition(order.begin(), order.end(), [&](uint32_t x) {
                const int64_t h = pairs[back[x]].reserved;
                return !(h > 0 && h * 4 < med);
            });
        }
    }
    // trace variant: the pairs of the query-profile kernel first, the comparing kernel's after them
    std::stable_partition(order.begin(), order.end(), [&](uint32_t x) { return dp[x].pad != 0; });
    const char* ev = getenv("IOC_ALIGN_VARIANT");
    const bool carry = ev && strcmp(ev, "carry") == 0;
    // IOC_ALIGN_PACKED=1: the query-profile pairs go through the packed 16-bit kernel (two row bands per wave).
    // Bit-identical and 30 % fewer VALU instructions, but no faster yet on MI355X: opt-in.
    const bool packed = !carry && !g_force32 && getenv("IOC_ALIGN_PACKED") && !getenv("IOC_ALIGN_NO_PACKED");
    const uint32_t colsper = carry ? ALN_C : FW_C;
    // waves per pair: few pairs -> wide workgroups (latency), many pairs -> narrow ones (no fill/drain waste)
    uint32_t waves = ALN_MAXW;
    if (const char* e = getenv("IOC_ALIGN_WAVES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= ALN_MAXW) waves = uint32_t(v);
    } else {
        while (waves > 1 && uint64_t(np) * waves > 4096) waves >>= 1;
        if (carry) {
            while (waves > 1 && uint64_t(waves / 2) * 64 * colsper >= max_m) waves >>= 1;
        } else {
            // row bands x column strips run as a pipeline: it needs a few strips per band to fill, and
            // whole tiles per band
            const uint32_t strips = (max_m + 64 * FW_C - 1) / (64 * FW_C), tiles = (max_n + TILE - 1) / TILE;
            while (waves > 1 && (strips < 2 * waves || tiles < 2 * waves)) waves >>= 1;
            if (packed) while (waves > 1 && (strips < 4 * waves || tiles < 4 * waves)) waves >>= 1;  // 2 bands per wave
        }
    }
    if (packed && waves > uint32_t(P16_MAXW)) waves = P16_MAXW;
    // a handful of long pairs (the later alignment rounds of a batch): 4-wave workgroups, several per pair (below)
    const bool few_pairs = !carry && !packed && !getenv("IOC_ALIGN_WAVES") && np <= 64 && waves >= 4 && !getenv("IOC_ALIGN_NO_CROSS_TAIL");
    if (few_pairs) waves = 4;
    const uint32_t NT = waves * 64;
    int r;
    if ((r = reserve(c, c->a_pairs, size_t(np) * sizeof(AlnPairDev))) != IOC_OK) return r;
    if ((r = reserve(c, c->a_order, size_t(np) * 4)) != IOC_OK) return r;
    if ((r = reserve(c, c->a_out, size_t(np) * 8)) != IOC_OK) return r;
    hipStream_t s = c->stream;
    ACHK(c, hipMemcpyAsync(c->a_pairs.p, dp.data(), size_t(np) * sizeof(AlnPairDev), hipMemcpyHostToDevice, s));
    ACHK(c, hipMemcpyAsync(c->a_order.p, order.data(), size_t(np) * 4, hipMemcpyHostToDevice, s));
    AlnParams P{match, mismatch, gap_extend, uint32_t(k), 1u << (32 - k)};
    int32_t* d_score = static_cast<int32_t*>(c->a_out.p);
    uint32_t* d_count = reinterpret_cast<uint32_t*>(d_score + np);
    // Version 2 (ioc_align_v2.inc) takes the query-profile pairs — the front of `order`: two pairs per wave, tiles scheduled
    // by dataflow, coarse checkpoints.  IOC_ALIGN_V1=1 keeps everything on the first version (which also serves pairs with
    // other letters, pairs the 16-bit window refuses, and the whole batch should a bounded wait of version 2 run out).
    // IOC_ALIGN_ARENA=fat: a long-lived process that can spare the HBM (35 MB per 16.7 kb pair, 56 GB for config 3's batch,
    // allocated once and kept) takes version 1 — fine checkpoints straight from the forward pass, a traceback half as long,
    // ~6 % less time per batch.  The default is the lean version: a `cluster` process that starts by allocating tens of GB
    // pays seconds for the driver to hand them out.
    uint32_t n_v2 = 0;
    const char* arena_mode = getenv("IOC_ALIGN_ARENA");
    const bool v2 = !carry && !packed && !g_force32 && !getenv("IOC_ALIGN_V1") && !(arena_mode && strcmp(arena_mode, "fat") == 0);
    if (v2) {
        while (n_v2 < np && dp[order[n_v2]].pad != 0) ++n_v2;
        bool fell_back = false;
        if ((r = align_v2_run(c, dp, order.data(), n_v2, static_cast<const uint32_t*>(c->a_order.p), P, d_score, d_count, &fell_back)) != IOC_OK) return r;
        if (fell_back) {
            c->tm.n_align_refused += n_v2;
            n_v2 = 0;
        }
    }
    if (carry) {
        const uint64_t bnd_stride = 2ull * 6ull * max_n;
        const uint64_t lrow_entries = uint64_t((max_m + NT * ALN_C - 1) / (NT * ALN_C)) * NT;
        const uint64_t lrow_stride = lrow_entries * 4ull;
        // scratch is per workgroup; run in slices when the whole batch would not fit the budget
        const uint64_t budget = 8ull << 30;
        const uint64_t per_pair = (bnd_stride + lrow_stride) * 4ull;
        const uint32_t slice = uint32_t(std::min<uint64_t>(np, std::max<uint64_t>(1, budget / per_pair)));
        if ((r = reserve(c, c->a_bnd, size_t(slice) * bnd_stride * 4)) != IOC_OK) return r;
        if ((r = reserve(c, c->a_lrow, size_t(slice) * lrow_stride * 4)) != IOC_OK) return r;
        for (uint32_t first = 0; first < np; first += slice) {
            const uint32_t cnt = std::min(slice, np - first);
            hipLaunchKernelGGL(k_align_carry, dim3(cnt), dim3(NT), 0, s, static_cast<const AlnPairDev*>(c->a_pairs.p),
                               static_cast<const uint32_t*>(c->a_order.p) + first, static_cast<const uint8_t*>(c->a_pool.p), P,
                               static_cast<uint32_t*>(c->a_bnd.p), bnd_stride, static_cast<uint32_t*>(c->a_lrow.p),
                               lrow_stride, d_score, d_count);
            ACHK(c, hipGetLastError());
        }
    } else if (n_v2 < np) {
        // checkpoint arena: slices of pairs (in `order`) whose checkpoints fit the budget together
        size_t free_b = 0, total_b = 0;
        ACHK(c, hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = uint64_t(free_b + c->a_ck.cap) / 2;
        if (const char* e = getenv("IOC_ALIGN_CK_BUDGET_MB")) budget = uint64_t(atoll(e)) << 20;
        // (row region: the larger of the two forward kernels' formats, so that a refused pair can be re-run in place)
        auto row_units = [&](const AlnPairDev& d) {
            const uint64_t plain = uint64_t((d.n - 1) / TILE) * row_pitch(d.m);
            return (packed && d.pad) ? std::max(plain, (row16_units(d.n, d.m) + 15u) & ~uint64_t(15)) : plain;
        };
        auto ck_units = [&](const AlnPairDev& d) { return row_units(d) + uint64_t((d.m - 1) / TILE) * col_pitch(d.n); };
        const uint64_t lrow_stride = uint64_t((max_m + 64 * FW_C - 1) / (64 * FW_C)) * 64;  // int2 per pair
        // (the forward pass addresses a pair's row checkpoints by a 32-bit offset in ints: 2^29 int2 units = 4 GB per pair,
        // reads of ~260 kb against each other)
        for (uint32_t x = 0; x < np; ++x)
            if (row_units(dp[x]) >= (uint64_t(1) << 29))
                return ioc_fail(c, IOC_ERR_CAPACITY, "alignment of " + std::to_string(dp[x].n) + " x " + std::to_string(dp[x].m) +
                                                        " bases: row checkpoints above 4 GB per pair");
        std::vector<AlnCk> cko(np);
        std::vector<std::pair<uint32_t, uint32_t>> slices;  // [first, count) in `order`
        uint64_t arena = 0;
        for (uint32_t first = n_v2; first < np;) {
            uint64_t used = 0;
            uint32_t cnt = 0;
            while (first + cnt < np) {
                const AlnPairDev& d = dp[order[first + cnt]];
                const uint64_t u = ck_units(d);
                if (cnt > 0 && (used + u) * 8ull > budget) break;
                if (cnt > 0 && d.pad != dp[order[first]].pad) break;  // one kernel per slice
                // (rows first: their pitch is a multiple of 16 int2, `used` stays 16-byte aligned for the int4 stores)
                cko[order[first + cnt]] = AlnCk{used, used + row_units(d)};
                used += u;
                ++cnt;
            }
            arena = std::max(arena, used);
            slices.emplace_back(first, cnt);
            first += cnt;
        }
        uint32_t max_cnt = 0;
        for (auto& sl : slices) max_cnt = std::max(max_cnt, sl.second);
        const auto t_res0 = std::chrono::steady_clock::now();
        if ((r = reserve(c, c->a_ck, size_t(arena) * 8)) != IOC_OK) return r;
        c->tm.align_arena_bytes = int64_t(arena) * 8;
        c->tm.align_slices = int32_t(slices.size());
        c->tm.align_version = 1;
        if (getenv("IOC_TRACE"))
            fprintf(stderr, "[ioc]   aligner: checkpoint arena %.1f MB (%zu slice(s)) reserved in %.3f ms\n", double(arena) * 8e-6, slices.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_res0).count());
        if ((r = reserve(c, c->a_cko, size_t(np) * sizeof(AlnCk))) != IOC_OK) return r;
        if ((r = reserve(c, c->a_ends2, size_t(np) * sizeof(int4))) != IOC_OK) return r;
        if ((r = reserve(c, c->a_lrow, size_t(max_cnt) * lrow_stride * 8)) != IOC_OK) return r;
        ACHK(c, hipMemcpyAsync(c->a_cko.p, cko.data(), size_t(np) * sizeof(AlnCk), hipMemcpyHostToDevice, s));
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        if (n_cu < 1) n_cu = 256;
        std::vector<hipEvent_t> evs(slices.size() * 3, nullptr);
        for (auto& e : evs) ACHK(c, hipEventCreate(&e));
        size_t evi = 0;
        for (auto& sl : slices) {
            const uint32_t* ord = static_cast<const uint32_t*>(c->a_order.p) + sl.first;
            // Every workgroup lives for the whole launch (equal-sized pairs), so the launch ends with the
            // fullest CU: cap the residency at ceil(workgroups / CUs) per CU with an LDS reservation, or the
            // dispatcher may stack 8 workgroups on some CUs and leave others with 3.
            const bool prof = dp[order[sl.first]].pad != 0;
            const void* kfn = prof ? (packed ? reinterpret_cast<const void*>(k_align_fwd16) : reinterpret_cast<const void*>(k_align_fwd<true>))
                                   : reinterpret_cast<const void*>(k_align_fwd<false>);
            size_t lds_pad = 0;
            const uint32_t wg_waves = waves > 4 ? waves : 4, ppw = wg_waves / waves;
            uint32_t n_wg = (sl.second + ppw - 1) / ppw, n_main = n_wg, wpp_tail = waves;
            {   // a second, partly filled generation of workgroups: give its pairs twice the waves
                int occ = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, int(wg_waves * 64), 0) != hipSuccess) occ = 0;
                (void)hipGetLastError();
                const uint32_t gen = uint32_t(occ > 0 ? occ : 3) * uint32_t(n_cu);  // resident workgroups
                if (!getenv("IOC_ALIGN_NO_TAIL") && n_wg > gen && n_wg < gen + gen / 2 && 2 * waves <= wg_waves) {
                    n_main = gen;
                    wpp_tail = 2 * waves;
                    const uint32_t rest = sl.second - n_main * ppw, ppw_tail = wg_waves / wpp_tail;
                    n_wg = n_main + (rest + ppw_tail - 1) / ppw_tail;
                }
            }
            const uint32_t per_cu = (n_main + uint32_t(n_cu) - 1) / uint32_t(n_cu);
            if (!getenv("IOC_ALIGN_NO_CAP") && per_cu * wg_waves < 32) {
                // per-workgroup LDS (static + reservation) halfway between 160 KB / (per_cu + 1) and 160 KB / per_cu
                hipFuncAttributes fa{};
                size_t stat = 48 * 1024;
                if (hipFuncGetAttributes(&fa, kfn) == hipSuccess) stat = fa.sharedSizeBytes;
                const size_t want = size_t(160u * 1024u) * (2 * per_cu + 1) / (2 * per_cu * (per_cu + 1));
                lds_pad = want > stat + 1024 ? (want - stat - 512) & ~size_t(255) : 0;
                // more than the default 64 KB per workgroup needs the kernel attribute (best effort)
                size_t& lim = prof ? (packed ? c->aln_lds_max3 : c->aln_lds_max) : c->aln_lds_max2;
                if (lim == 0) {
                    int mx = 0;
                    (void)hipDeviceGetAttribute(&mx, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device);
                    lim = 64 * 1024 > stat ? 64 * 1024 - stat : 1;
                    if (mx > 72 * 1024 && hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, mx - int(stat)) == hipSuccess)
                        lim = size_t(mx) - stat;
                    (void)hipGetLastError();
                }
                lds_pad = std::min(lds_pad, lim > 1 ? lim : 0);
            }
            ACHK(c, hipEventRecord(evs[evi++], s));
            auto launch_fwd = [&](const uint32_t* o, uint32_t cnt, uint32_t wv, uint32_t wgw, uint32_t nwg, uint32_t nmain,
                                  uint32_t wtail, size_t lds, int2* lr, const AlnCross& X) {
                const AlnPairDev* dpairs = static_cast<const AlnPairDev*>(c->a_pairs.p);
                const uint8_t* dpool = static_cast<const uint8_t*>(c->a_pool.p);
                int2* dck = static_cast<int2*>(c->a_ck.p);
                const AlnCk* dcko = static_cast<const AlnCk*>(c->a_cko.p);
                int4* dends = static_cast<int4*>(c->a_ends2.p);
                if (prof && packed)
                    hipLaunchKernelGGL(k_align_fwd16, dim3(nwg), dim3(wgw * 64), lds, s, dpairs, o, cnt, wv, nmain, wtail, dpool, P, dck,
                                       dcko, lr, lrow_stride, dends);
                else if (prof)
                    hipLaunchKernelGGL(k_align_fwd<true>, dim3(nwg), dim3(wgw * 64), lds, s, dpairs, o, cnt, wv, nmain, wtail, dpool, P,
                                       dck, dcko, lr, lrow_stride, dends, X);
                else
                    hipLaunchKernelGGL(k_align_fwd<false>, dim3(nwg), dim3(wgw * 64), lds, s, dpairs, o, cnt, wv, nmain, wtail, dpool, P,
                                       dck, dcko, lr, lrow_stride, dends, X);
            };
            // The tail generation runs on a mostly empty chip: its pairs are split over SEVERAL workgroups each
            // (4 x 4 = 16 bands on 4 CUs; an 8-wave workgroup lands on one CU, two waves per SIMD, and was slower),
            // the first band of a workgroup waiting on a flag for the row checkpoints of the band above.  Its own
            // launch: all its workgroups are resident together, and the waits are bounded anyway (then the pairs
            // are simply done again the usual way).
            uint32_t groups = 1;
            // the whole slice the tail's way: forced (tests), or a handful of pairs on an otherwise idle chip
            const bool force_cross = getenv("IOC_ALIGN_FORCE_CROSS") != nullptr || (few_pairs && waves == 4);
            if (force_cross && !packed && wg_waves == 4) {
                n_main = 0;
                n_wg = 1;
            }
            // Default: in the SAME launch, behind the ordinary workgroups, so that a split pair starts on whatever CU
            // frees up first (config 3: 105.2 -> 96.0 ms).  IOC_ALIGN_CROSS_TAIL=separate makes it a launch of its own
            // (one workgroup per CU by an LDS reservation; measured slower: it cannot backfill, 112 ms),
            // IOC_ALIGN_NO_CROSS_TAIL=1 keeps the tail pairs whole with twice the waves (the first version).
            // Workgroups of a launch are handed out in index order and a workgroup only waits for a smaller index;
            // the split ones are at most two thirds of an XCD's slots, so they can never fill an XCD with waiters.
            const char* ct0 = getenv("IOC_ALIGN_CROSS_TAIL");
            const bool inl = !(ct0 && strcmp(ct0, "separate") == 0) && !force_cross;
            if (n_main < n_wg && !packed && wg_waves == 4 && !getenv("IOC_ALIGN_NO_CROSS_TAIL")) {
                const AlnPairDev& big = dp[order[sl.first]];
                const uint32_t strips = (big.m + 64 * FW_C - 1) / (64 * FW_C), tiles = (big.n + TILE - 1) / TILE;
                groups = force_cross ? 4 : 2;  // (behind a full generation 4 was measured slower: 99 ms)
                if (const char* eg = getenv("IOC_ALIGN_CROSS_GROUPS")) groups = uint32_t(std::max(1, std::min(4, atoi(eg))));
                while (groups > 1 && (strips < 4 * groups || tiles < 8 * groups)) groups >>= 1;  // bands = 4 * groups
                const uint32_t rest = sl.second - n_main * ppw;
                // one workgroup per CU (two on a CU run at half speed and hold up the whole chain of their pair), all
                // resident together
                // (in-launch: at most 2 split workgroups per CU's worth, i.e. 64 of an XCD's 96 slots — waiters cannot
                // fill an XCD; a launch of its own: one per CU)
                while (groups > 1 && uint64_t(rest) * groups > uint64_t(inl ? 2 : 1) * uint64_t(n_cu)) groups >>= 1;
            }
            if (groups > 1) {
                const uint32_t first_cnt = n_main * ppw, rest = sl.second - first_cnt, nb = 4 * groups;
                // flags are indexed [band][strip] with every pair's OWN strip count: a pair with fewer cells than the
                // slice's first one may still have the longer reference, so the slot holds the largest count of the
                // pairs that are actually split
                uint32_t strips = 1;
                for (uint32_t x = first_cnt; x < sl.second; ++x)
                    strips = std::max(strips, (dp[order[sl.first + x]].m + 64 * FW_C - 1) / (64 * FW_C));
                AlnCross X;
                X.groups = groups;
                X.flag_stride = nb * strips;
                const size_t fbytes = size_t(rest) * X.flag_stride * 4, bbytes = size_t(rest) * nb * sizeof(int2);
                if ((r = reserve(c, c->a_xflags, fbytes + bbytes + 64)) != IOC_OK) return r;
                X.err = static_cast<uint32_t*>(c->a_xflags.p);
                X.flags = X.err + 16;
                X.best = reinterpret_cast<int2*>(reinterpret_cast<uint8_t*>(c->a_xflags.p) + 64 + fbytes);
                ACHK(c, hipMemsetAsync(c->a_xflags.p, 0, 64 + fbytes, s));
                if (inl && first_cnt) {
                    // one launch: ordinary workgroups, then the split pairs behind them
                    X.first_wg = n_main;
                    X.first_pair = first_cnt;
                    launch_fwd(ord, sl.second, waves, wg_waves, n_main + rest * groups, n_main + rest * groups, waves, lds_pad,
                               static_cast<int2*>(c->a_lrow.p), X);
                    ACHK(c, hipGetLastError());
                    uint32_t xe = 0;
                    ACHK(c, hipMemcpyAsync(&xe, X.err, 4, hipMemcpyDeviceToHost, s));
                    ACHK(c, hipStreamSynchronize(s));
                    if (xe) {
                        launch_fwd(ord + first_cnt, rest, 4, 4, rest, rest, 4, 0, static_cast<int2*>(c->a_lrow.p) + uint64_t(first_cnt) * lrow_stride, AlnCross{});
                        c->tm.n_align_refused += rest;
                    }
                } else {
                if (first_cnt) launch_fwd(ord, first_cnt, waves, wg_waves, n_main, n_main, waves, lds_pad, static_cast<int2*>(c->a_lrow.p), AlnCross{});
                ACHK(c, hipGetLastError());
                size_t xlds = 0;  // more than half a CU's LDS: the dispatcher cannot stack two of them
                {
                    hipFuncAttributes fa{};
                    size_t stat = 52 * 1024;
                    if (hipFuncGetAttributes(&fa, kfn) == hipSuccess) stat = fa.sharedSizeBytes;
                    size_t& lim = prof ? c->aln_lds_max : c->aln_lds_max2;
                    if (lim == 0) {
                        int mx = 0;
                        (void)hipDeviceGetAttribute(&mx, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device);
                        lim = 64 * 1024 > stat ? 64 * 1024 - stat : 1;
                        if (mx > 72 * 1024 && hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, mx - int(stat)) == hipSuccess)
                            lim = size_t(mx) - stat;
                        (void)hipGetLastError();
                    }
                    const size_t want = 84 * 1024;
                    if (want > stat && want - stat <= lim) xlds = (want - stat) & ~size_t(255);
                }
                launch_fwd(ord + first_cnt, rest, 4, 4, rest * groups, rest * groups, 4, xlds,
                           static_cast<int2*>(c->a_lrow.p) + uint64_t(first_cnt) * lrow_stride, X);
                ACHK(c, hipGetLastError());
                uint32_t xerr = 0;
                ACHK(c, hipMemcpyAsync(&xerr, X.err, 4, hipMemcpyDeviceToHost, s));
                ACHK(c, hipStreamSynchronize(s));
                if (xerr) {  // a wait ran out (the workgroups were not all resident): the tail again, the usual way
                    launch_fwd(ord + first_cnt, rest, 4, 4, rest, rest, 4, 0,
                               static_cast<int2*>(c->a_lrow.p) + uint64_t(first_cnt) * lrow_stride, AlnCross{});
                    c->tm.n_align_refused += rest;
                }
                }
            } else {
                if (force_cross && n_main == 0) {  // (the forced split did not fit this slice)
                    n_wg = (sl.second + ppw - 1) / ppw;
                    n_main = n_wg;
                }
                launch_fwd(ord, sl.second, waves, wg_waves, n_wg, n_main, wpp_tail, lds_pad, static_cast<int2*>(c->a_lrow.p), AlnCross{});
            }
            ACHK(c, hipGetLastError());
            ACHK(c, hipEventRecord(evs[evi++], s));
            const uint32_t tr_wg = (sl.second + TR_WAVES - 1) / TR_WAVES;  // (an LDS reservation that caps the workgroups per CU changed nothing here)
            hipLaunchKernelGGL(k_align_trace, dim3(tr_wg), dim3(64 * TR_WAVES), 0, s,
                               static_cast<const AlnPairDev*>(c->a_pairs.p), ord, static_cast<const uint8_t*>(c->a_pool.p), P,
                               static_cast<const int2*>(c->a_ck.p), static_cast<const AlnCk*>(c->a_cko.p),
                               static_cast<const int4*>(c->a_ends2.p), d_score, d_count, sl.second);
            ACHK(c, hipGetLastError());
            ACHK(c, hipEventRecord(evs[evi++], s));
        }
        ACHK(c, hipStreamSynchronize(s));
        for (size_t x = 0; x + 2 < evs.size() + 0 && x < evi; x += 3) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, evs[x], evs[x + 1]) == hipSuccess) c->tm.ms_align_fwd += a;
            if (hipEventElapsedTime(&b, evs[x + 1], evs[x + 2]) == hipSuccess) c->tm.ms_align_trace += b;
        }
        if (getenv("IOC_TRACE"))
            fprintf(stderr, "[ioc]   aligner: forward %.3f ms, traceback %.3f ms (device, all slices so far)\n", c->tm.ms_align_fwd, c->tm.ms_align_trace);
        for (auto& e : evs) (void)hipEventDestroy(e);
    }
    c->tm.n_align_pairs += np;
    for (auto& d : dp) c->tm.n_align_cells += int64_t(d.n) * int64_t(d.m);
    for (uint32_t x = n_v2; x < np; ++x) c->tm.n_align_cells_computed += int64_t(dp[order[x]].n) * int64_t(dp[order[x]].m);  // (version 1: every cell; version 2 counts its tiles)
    std::vector<int32_t> hs(np);
    std::vector<uint32_t> hc(np);
    ACHK(c, hipMemcpyAsync(hs.data(), d_score, size_t(np) * 4, hipMemcpyDeviceToHost, s));
    ACHK(c, hipMemcpyAsync(hc.data(), d_count, size_t(np) * 4, hipMemcpyDeviceToHost, s));
    ACHK(c, hipStreamSynchronize(s));
    if (v2 && !g_no_corridor) {
        // what the corridor model learns from this batch (ioc_ctx::aln_fit): related pairs long enough to have had a corridor
        if (c->aln_fit_sig[0] != match || c->aln_fit_sig[1] != mismatch || c->aln_fit_sig[2] != gap_extend) {
            for (auto& f : c->aln_fit) f = ioc_ctx::CorridorFit();
            c->aln_fit_sig[0] = match, c->aln_fit_sig[1] = mismatch, c->aln_fit_sig[2] = gap_extend;
        }
        for (uint32_t x = 0; x < np; ++x) {
            const AlnPairDev& d = dp[x];
            const uint32_t len = std::max(d.n, d.m);
            if (hc[x] == 0xFFFFFFFFu || len < 4096u || !(d.e_sum > 0.0f) || std::min(d.n, d.m) * 10u < len * 9u) continue;
            const double rho = double(hs[x]) / double(len), e = double(d.e_sum);
            if (rho * 2.0 < double(match)) continue;  // (not a pair of one transcript)
            ioc_ctx::CorridorFit& f = c->aln_fit[std::max(2, std::min(5, d.gap_open)) - 2];
            if (f.n >= 1e6) continue;  // (enough)
            f.n += 1.0, f.se += e, f.sr += rho, f.see += e * e, f.ser += e * rho, f.srr += rho * rho;
            f.e_lo = std::min(f.e_lo, e), f.e_hi = std::max(f.e_hi, e);
        }
    }
    std::vector<int32_t> again;  // pairs the packed kernel flagged: scores outside its 16-bit window
    std::vector<int32_t> again_full;  // pairs whose result the corridor cannot vouch for (V2Couple): every tile this time
    for (uint32_t x = 0; x < np; ++x) {
        const uint32_t i = back[x];
        if ((packed || v2) && hc[x] == 0xFFFFFFFFu && hs[x] == INT32_MIN) {
            again.push_back(int32_t(i));
            continue;
        }
        if (v2 && hc[x] == 0xFFFFFFFFu && hs[x] == INT32_MIN + 1) {
            again_full.push_back(int32_t(i));
            continue;
        }
        if (out_score) out_score[i] = hs[x];
        if (out_windows) out_windows[i] = int64_t(hc[x]);
        if (out_ratio) out_ratio[i] = double(hc[x]) / double(dp[x].n);  // getAlnRatio: aligned / slen
    }
    if (!again_full.empty()) {
        std::vector<ioc_aln_pair> sub(again_full.size());
        std::vector<int32_t> sc(again_full.size());
        std::vector<int64_t> sw(again_full.size());
        std::vector<double> sr(again_full.size());
        for (size_t x = 0; x < again_full.size(); ++x) sub[x] = pairs[again_full[x]];
        const int64_t pairs_so_far = c->tm.n_align_pairs, cells_so_far = c->tm.n_align_cells;  // (a pair counts once)
        g_no_corridor = true;
        const int rr = ioc_align_pairs(c, int32_t(sub.size()), sub.data(), k, match, mismatch, gap_extend, sc.data(), sw.data(), sr.data());
        g_no_corridor = false;
        if (rr != IOC_OK) return rr;
        c->tm.n_align_pairs = pairs_so_far;
        c->tm.n_align_cells = cells_so_far;
        if (getenv("IOC_TRACE")) fprintf(stderr, "[ioc]   aligner v2: %zu of %u pairs run again without a corridor\n", again_full.size(), np);
        for (size_t x = 0; x < again_full.size(); ++x) {
            if (out_score) out_score[again_full[x]] = sc[x];
            if (out_windows) out_windows[again_full[x]] = sw[x];
            if (out_ratio) out_ratio[again_full[x]] = sr[x];
        }
    }
    if (!again.empty()) {
        std::vector<ioc_aln_pair> sub(again.size());
        std::vector<int32_t> sc(again.size());
        std::vector<int64_t> sw(again.size());
        std::vector<double> sr(again.size());
        for (size_t x = 0; x < again.size(); ++x) sub[x] = pairs[again[x]];
        g_force32 = true;
        const int rr = ioc_align_pairs(c, int32_t(sub.size()), sub.data(), k, match, mismatch, gap_extend, sc.data(), sw.data(), sr.data());
        g_force32 = false;
        if (rr != IOC_OK) return rr;
        c->tm.n_align_refused += int64_t(again.size());
        for (size_t x = 0; x < again.size(); ++x) {
            if (out_score) out_score[again[x]] = sc[x];
            if (out_windows) out_windows[again[x]] = sw[x];
            if (out_ratio) out_ratio[again[x]] = sr[x];
        }
    }
    return IOC_OK;
}

// (ioc_ctx_prewarm)
hipError_t iock_warm_align()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_fwd2_ends));
}

}  // extern "C"
