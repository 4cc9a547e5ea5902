// ioc_poa.hip — partial-order alignment for the consensus of a cluster.
//
// The reference keeps one spoa::Graph per cluster and performs five operations on it (src/consensus.cpp:15-32,
// 41, 87, 128-137; src/cluster.cpp:200-204): seed, size, AddSeqToGraph (align + AddAlignment), GenerateConsensus,
// ConsPurge, with spoa::AlignmentEngine::Create(kSW, m=4, n=-8, g=-8, e=-4, q=-20, c=-1) (src/main.cpp:285-324).
// spoa is a third-party library ABSENT from /root/reference (.gitmodules:13-15): this is an engine of its own
// after the published algorithm (Lee, Grasso & Sharlow 2002: sequence-to-graph DP over the topologically sorted
// nodes; convex gaps as two affine pieces; heaviest-bundle consensus) — PARITY WITH spoa IS UNPINNED:
// alignments of equal score and the consensus' tie-breaks may differ.
//
// What runs where: the graphs (nodes, weighted edges, aligned-node groups, topological order) and the consensus
// live on the host — they are small.  The sequence-to-graph DP is the heavy part (|graph| x |read| cells, 3*10^8
// for a 16.7 kb read) and runs on the GPU:
//   k_poa_tile     rows = graph nodes in topological order.  The columns of a row are independent once its
//                  predecessor rows are known, except for the horizontal gap states, which are a max-plus
//                  prefix scan along the row: E[j] = max_{x<j} Hn[x] + open + (j-1-x) ext, Hn = H without E (opening
//                  a gap right after a gap never beats extending it, and switching between the two pieces of the
//                  convex gap never pays when open <= ext for both pieces, which is checked).  The matrix is cut
//                  into tiles of 64 rows x 256 columns (one column per thread) and swept one anti-diagonal of
//                  tiles per launch: tiles of a diagonal are independent, a tile takes its row carries and
//                  its boundary column from the tile on its left and keeps its last 16 rows in LDS.  Stores
//                  H, F1, F2 (later rows need them), a direction word and an E byte per cell.
//   k_poa_trace    one wave walks back from the best cell over the direction words, 64 cells of look-ahead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "ioc_internal.h"

namespace {

constexpr int POA_NEG = INT32_MIN / 4;
#ifndef IOC_POA_THREADS
#define IOC_POA_THREADS 256
#endif
constexpr int POA_THREADS = IOC_POA_THREADS;  // columns of a tile = threads of its workgroup (a multiple of 64)
constexpr int POA_MAX_COLS = 1 << 20;
constexpr int POA_MAX_PREDS = 127;

// direction word: [1:0] source of Hn (0 stop, 1 diagonal, 2 F1, 3 F2)  [7] F1 extended  [8] F2 extended
// [2] / [3]: extending and opening F1 / F2 score the same (the traceback's rule for that tie differs between the step that
// ENTERS a vertical gap from H — extension first — and the steps that follow the gap up — opening first: DESIGN.md 5.7)
// (a horizontal gap winning H is in the cell's E byte: [2:0] 4 E1 / 5 E2 / 0, [3] E1 extended, [4] E2 extended; a tie
// between extending and opening a horizontal gap counts as extended)
// [15:9] / [22:16] / [29:23] predecessor (index into the row's predecessor list) of the diagonal / F1 / F2 move
enum : uint32_t { SRC_STOP = 0, SRC_DIAG = 1, SRC_F1 = 2, SRC_F2 = 3, SRC_E1 = 4, SRC_E2 = 5 };

struct PoaScores {
    int m, n, g, e, q, c;
};

constexpr int POA_CB = POA_THREADS;  // columns per tile: one per thread
constexpr int POA_RB = 64;           // rows per tile
constexpr int POA_WAVES = POA_THREADS / 64;
#ifndef IOC_POA_RING
#define IOC_POA_RING 8
#endif
constexpr int POA_RING = IOC_POA_RING;  // rows of the tile kept in LDS behind the current one (a power of two)
constexpr int POA_PRED_LDS = 448;    // predecessor entries of a tile staged in LDS (the rest is read from memory)

// One alignment of a batch (blockIdx.y): a read against one graph.  Graphs are independent, so the pending
// additions of MANY clusters are aligned by the same launches (a single alignment offers at most 66 tiles per
// diagonal to 256 CUs).
struct PoaJob {
    int R, L, nrb, ncb;
    const uint8_t* base;
    const int32_t* pred_off;
    const int32_t* pred;
    const int32_t* pred_slot;  // per predecessor entry: the plane row of that predecessor's H / F1 / F2, -1 if not kept
    const int32_t* slot;       // per row: its plane row, -1 if no later row reads it from memory
    const uint8_t* seq;
    int* H;
    int* F1;
    int* F2;
    uint32_t* dirs;
    uint8_t* ebits;
    int4* carry;
    int4* tile_best;
    int* best;  // score, row, column
    int32_t* out_node;
    int32_t* out_pos;
    int32_t* out_n;
    int cap;
};

// row 0 = virtual source (H = 0: local alignment)
__global__ void k_poa_init(const PoaJob* __restrict__ jobs)
{
    const PoaJob J = jobs[blockIdx.y];
    const int W = J.L + 1;
    int *H = J.H, *F1 = J.F1, *F2 = J.F2;
    uint32_t* dirs = J.dirs;
    uint8_t* ebits = J.ebits;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < W) {
        H[j] = 0;
        F1[j] = POA_NEG;
        F2[j] = POA_NEG;
        dirs[j] = SRC_STOP;
        ebits[j] = 0;
    }
}

using GI32 = __attribute__((address_space(1))) int32_t;
using LI32 = __attribute__((address_space(3))) int32_t;  // (LDS: a generic pointer would make the access a FLAT one)
using GU32 = __attribute__((address_space(1))) uint32_t;
using GU8 = __attribute__((address_space(1))) uint8_t;

// lanes without a source (and rows masked off) receive INT32_MIN, THE identity of a signed maximum: the compiler folds such
// a move into the maximum that consumes it (v_max_i32_dpp, one instruction per stage instead of identity + move + maximum;
// with POA_NEG in its place it does not).  The value only ever meets max(), never an addition.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_or_self(int v)
{
    return __builtin_amdgcn_update_dpp(INT32_MIN, v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive prefix maxima over the 64 lanes of a wave, two values at once: row_shr 1 2 4 8, then row_bcast 15 and 31
__device__ __forceinline__ void wave_prefix_max2(int& a, int& b)
{
    a = max(a, dpp_or_self<0x111>(a));
    b = max(b, dpp_or_self<0x111>(b));
    a = max(a, dpp_or_self<0x112>(a));
    b = max(b, dpp_or_self<0x112>(b));
    a = max(a, dpp_or_self<0x114>(a));
    b = max(b, dpp_or_self<0x114>(b));
    a = max(a, dpp_or_self<0x118>(a));
    b = max(b, dpp_or_self<0x118>(b));
    a = max(a, dpp_or_self<0x142, 0xa>(a));
    b = max(b, dpp_or_self<0x142, 0xa>(b));
    a = max(a, dpp_or_self<0x143, 0xc>(a));
    b = max(b, dpp_or_self<0x143, 0xc>(b));
}

// One anti-diagonal of tiles per launch (tile = POA_RB rows x POA_CB columns): the tiles of a diagonal are
// independent, kernel boundaries are the only synchronisation between tiles (no flags, no cross-workgroup
// coherence games).  Inside a tile the rows are sequential and latency is everything:
//   * the 4 waves of the workgroup own 64 columns each and run one row apart (wave w is on row s - w in step s),
//     so the only thing that crosses waves is the row carry of the wave on the left, written one step earlier:
//     one LDS barrier per step, and the row scan itself is DPP inside the wave;
//   * nothing a row needs comes from memory in the common case: the tile's predecessor lists, bases and left
//     carries are staged in LDS once, the last POA_RING rows of H / F1 / F2 stay in an LDS ring that only the
//     owning lane touches (a predecessor further back, or above the tile, is read from memory), and the next
//     row's list head is fetched while the current row computes.
// rows 1..R = nodes in topological order; pred_off[r] .. pred_off[r+1]: predecessor ROWS of row r (row 0 for a
// node without in-edges).  carry[cb][r] = (prefix max of Hn[x] - e x, of Hn[x] - c x, over all columns up to the
// tile's last one; Hn and H of that last column): what the tile to the right needs of row r.
__global__ void __launch_bounds__(POA_THREADS)
k_poa_tile(const PoaJob* __restrict__ jobs, int diag, PoaScores S, int pred_lds)
{
    __shared__ int sRing[3][POA_RING][POA_CB];  // H, F1, F2 (one array: one address register, the planes are immediate offsets)
#define sH sRing[0]
#define sF1 sRing[1]
#define sF2 sRing[2]
    // a row's carry into a wave — [0]: from the tile on the left; [w]: from wave w - 1 — as two 8-byte halves: (prefix maxima of the
    // two pieces) and (H of the last column, the strict-maximum flags | (row of the tile + 1) << 2).  The second half is written
    // AFTER the first and read BEFORE it (a wave's LDS operations are carried out in program order): a reader that finds the row's
    // tag has the whole carry — no separate "rows done" counter, one LDS round trip per row instead of two
    __shared__ __attribute__((aligned(16))) int4 s_carry[POA_WAVES][POA_RB];
    __shared__ int s_poff[POA_RB + 2];
    __shared__ int s_pred[POA_PRED_LDS], s_pslot[POA_PRED_LDS];
    __shared__ int s_base[POA_RB + 1], s_slot[POA_RB + 1];  // (one past the last row: read ahead, never used)
    __shared__ int s_red[3 * POA_WAVES];
    __shared__ int4 s_rec[POA_RB + 1];  // per row: end of its predecessor list, base, plane row, first predecessor — what a wave fetches one row ahead
    const PoaJob J = jobs[blockIdx.y];
    const int R = J.R, L = J.L, nrb = J.nrb;
    const int cb_first = max(0, diag - nrb + 1), cb_last = min(J.ncb - 1, diag);
    if (cb_first + int(blockIdx.x) > cb_last) return;  // this job has fewer tiles on the diagonal (or is finished)
    // (pointers out of the job record are generic to the compiler: say that they are global memory, or every
    // access becomes a FLAT one, which waits on the LDS and the memory counters together)
    const GI32* __restrict__ pred_off = (const GI32*)(J.pred_off);
    const GI32* __restrict__ pred = (const GI32*)(J.pred);
    const GI32* __restrict__ pred_slot = (const GI32*)(J.pred_slot);
    const GU8* __restrict__ seq = (const GU8*)(J.seq);
    GI32 *H = (GI32*)(J.H), *F1 = (GI32*)(J.F1), *F2 = (GI32*)(J.F2);
    GU32* __restrict__ dirs = (GU32*)(J.dirs);
    GU8* __restrict__ ebits = (GU8*)(J.ebits);
    GI32* carry = (GI32*)(J.carry);  // 4 values per (column tile, row)
    GI32* __restrict__ tile_best = (GI32*)(J.tile_best);
    const int W = L + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = cb_first + blockIdx.x, rb = diag - cb;
    const int j = cb * POA_CB + tid;
    const bool active = j < W;
    const int jj = min(j, W - 1);  // threads past the last column run along and store nothing
    const uint32_t jb4 = uint32_t(jj) * 4u;  // (W <= 2^20: fits)
    const int ofs_e = active ? -S.e * j : POA_NEG, ofs_c = active ? -S.c * j : POA_NEG;  // (hn < 2^23: POA_NEG + hn stays an outsider)
    const int r_lo = rb * POA_RB + 1, r_hi = min(R, r_lo + POA_RB - 1);
    const GI32* cin_row = carry + int64_t(max(cb, 1) - 1) * (R + 1) * 4;
    GI32* cout_row = carry + int64_t(cb) * (R + 1) * 4;
    // ---- the tile's rows: predecessor lists, bases, carries of the tile on the left ----
    const int nrows = r_hi - r_lo + 1;
    const int pb0 = pred_off[r_lo];
    for (int t = tid; t < (POA_WAVES - 1) * POA_RB; t += POA_THREADS) s_carry[1 + t / POA_RB][t % POA_RB] = int4{0, 0, 0, 0};  // (no row's tag)
    for (int t = tid; t <= nrows; t += POA_THREADS) s_poff[t] = pred_off[r_lo + t];
    if (tid == 0) s_poff[nrows + 1] = 0;
    if (tid == 0) s_rec[nrows] = int4{0, 0, 0, 0};  // (read ahead by the last row, never used)
    for (int t = tid; t < nrows; t += POA_THREADS) {
        s_base[t] = ((const GU8*)J.base)[r_lo + t];
        s_slot[t] = ((const GI32*)J.slot)[r_lo + t];
        s_rec[t] = int4{pred_off[r_lo + t + 1], int(((const GU8*)J.base)[r_lo + t]), ((const GI32*)J.slot)[r_lo + t], pred[pred_off[r_lo + t]]};
        const GI32* ci = cin_row + int64_t(r_lo + t) * 4;
        s_carry[0][t] = cb > 0 ? int4{ci[0], ci[1], ci[3], ci[2] | ((t + 1) << 2)} : int4{POA_NEG, POA_NEG, 0, (t + 1) << 2};
    }
    {
        const int np = min(pred_off[r_hi + 1] - pb0, pred_lds);
        for (int t = tid; t < np; t += POA_THREADS) {
            s_pred[t] = pred[pb0 + t];
            s_pslot[t] = pred_slot[pb0 + t];
        }
    }
    int my_base = j > 0 ? seq[jj - 1] : 0;
    asm volatile("" : "+v"(my_base));  // loaded before the row loop starts: no memory-counter wait inside it
    __syncthreads();
    int my_best = 0, my_r = 0, my_j = 0;
    // what the wave's next row starts with, fetched one step ahead: list bounds, base, first predecessor
    int pb = __builtin_amdgcn_readfirstlane(s_poff[0]), pe = __builtin_amdgcn_readfirstlane(s_poff[1]), bs = s_rec[0].y, pr0 = s_rec[0].w;
    int my_slot = s_rec[0].z;
    int edge_prev = 0;  // H of the column left of this wave's first, one row up (the previous row's carry)
#ifdef POA_PROF
    unsigned long long tp[4] = {0, 0, 0, 0}, tp_t = __builtin_readcyclecounter();
#define POA_TICK(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tp[k] += t_ - tp_t; tp_t = t_; }
#else
#define POA_TICK(k)
#endif
    // The waves are a pipeline, not a phalanx: wave w needs of row t only the carry wave w - 1 wrote when IT finished row t, so
    // it waits for that (the row's tag in the carry itself) instead of meeting all four waves at a barrier after every row — a
    // third of a tile's time was spent at that barrier waiting for whichever wave the scheduler had served last — and only where
    // the carry is first needed: the row's vertical moves and its own prefix scan do not depend on the wave on the left.
    // Wave 0 never waits (its carries come from the tile on the left, tagged when they were staged), so every wait ends.
    const uint32_t carry_in = uint32_t(reinterpret_cast<uintptr_t>(&s_carry[wave][0]));  // LDS byte addresses
    const uint32_t carry_out = uint32_t(reinterpret_cast<uintptr_t>(&s_carry[min(wave + 1, POA_WAVES - 1)][0]));
    int ph = 0, p1 = POA_NEG, p2 = POA_NEG;  // this lane's H, F1, F2 of the row before (what a chain row reads: no LDS round trip)
    for (int t = 0; t < nrows; ++t) {
        {
            const int r = r_lo + t;
            const int4 nx = s_rec[t + 1];  // the next row (made uniform at the end of the step: no wait for it up here)
            const int pe_nv = nx.x, bs_n = nx.y, slot_n = nx.z, pr0_n = nx.w;
            int hn = 0, f1 = POA_NEG, f2 = POA_NEG;
            uint32_t d = SRC_STOP;
            {
                uint32_t pw = 0, f1x = 0, f2x = 0, ft = 0;  // pw: the predecessors the three moves came from (0 on a chain row); ft: the tie bits
                uint32_t f1p = 0, f2p = 0;
                int dg = POA_NEG;
                const int sc = (bs == my_base) ? S.m : S.n;
                const int pr0u = __builtin_amdgcn_readfirstlane(pr0);
                if (pe - pb == 1 && t > 0 && pr0u == r - 1) {
                    // the common row: ONE predecessor, the row above, in this tile — straight-line code (the general
                    // loop below spends more time in its branches than in its arithmetic); same values bit for bit
                    const int hu = ph, u1 = p1, u2 = p2;
                    const int hl = __builtin_amdgcn_update_dpp(edge_prev, hu, 0x138, 0xf, 0xf, false);  // wave_shr:1
                    // (H of a row is >= 0: hl + sc, hu + g and hu + q are far above POA_NEG, the general loop's "is it better
                    // than nothing" tests are true here)
                    if (j > 0) dg = hl + sc;
                    const int o1 = hu + S.g, x1 = u1 + S.e;
                    f1 = max(o1, x1);
                    f1x = x1 > o1 ? 1u : 0u;
                    const int o2 = hu + S.q, x2 = u2 + S.c;
                    f2 = max(o2, x2);
                    f2x = x2 > o2 ? 1u : 0u;
                    ft = (x1 == o1 ? 4u : 0u) | (x2 == o2 ? 8u : 0u);
                } else {
                uint32_t dp = 0;
                // plane row of predecessor entry x (only the memory paths ask: a row read from memory is a kept one)
                auto pslot_of = [&](int x) {
                    int v = s_pslot[min(x - pb0, pred_lds - 1)];
                    if (x - pb0 >= pred_lds) v = pred_slot[x];
                    return __builtin_amdgcn_readfirstlane(v);
                };
                for (int x = pb; x < pe; ++x) {
                    int prv = pr0;
                    if (x > pb) {
                        prv = s_pred[min(x - pb0, pred_lds - 1)];
                        if (x - pb0 >= pred_lds) {  // (a tile with very many edges)
                            prv = pred[x];
                            asm volatile("" : "+v"(prv));  // waited for here, not at the join (see below)
                        }
                    } else if (x - pb0 >= pred_lds) {
                        prv = pred[x];
                        asm volatile("" : "+v"(prv));
                    }
                    const int pr = __builtin_amdgcn_readfirstlane(prv);
                    int hu, u1, u2, hl;
                    if (pr >= r_lo) {
                        const int edge = s_carry[wave][pr - r_lo].z;  // H of the column left of this wave's first
                        if (r - pr <= POA_RING) {
                            const int sl = pr & (POA_RING - 1);
                            hu = sH[sl][tid];
                            u1 = sF1[sl][tid];
                            u2 = sF2[sl][tid];
                        } else {
                            // a row of this tile that left the ring: this wave's own stores, once they have landed,
                            // read past the L1 (lines of it may have come in before the stores)
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const int64_t po = int64_t(pslot_of(x)) * W + jj;
                            hu = __hip_atomic_load((int*)(H + po), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            u1 = __hip_atomic_load((int*)(F1 + po), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            u2 = __hip_atomic_load((int*)(F2 + po), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("" : "+v"(hu), "+v"(u1), "+v"(u2));
                        }
                        hl = __builtin_amdgcn_update_dpp(edge, hu, 0x138, 0xf, 0xf, false);  // wave_shr:1
                    } else {
                        const int64_t po = int64_t(pslot_of(x)) * W;
                        hu = H[po + jj];
                        u1 = F1[po + jj];
                        u2 = F2[po + jj];
                        hl = jj > 0 ? H[po + jj - 1] : 0;
                        // the loads are waited for HERE: at the join the compiler would wait for the memory counter
                        // on the LDS path too, and that counter also holds the previous rows' stores
                        asm volatile("" : "+v"(hu), "+v"(u1), "+v"(u2), "+v"(hl));
                    }
                    if (j > 0) {
                        const int hd = hl + sc;
                        if (hd > dg) {
                            dg = hd;
                            dp = uint32_t(x - pb);
                        }
                    }
                    const int o1 = hu + S.g, x1 = u1 + S.e;
                    const int v1 = max(o1, x1);
                    if (v1 > f1) {
                        f1 = v1;
                        f1p = uint32_t(x - pb);
                        f1x = x1 > o1 ? 1u : 0u;
                        ft = (ft & ~4u) | (x1 == o1 ? 4u : 0u);
                    }
                    const int o2 = hu + S.q, x2 = u2 + S.c;
                    const int v2 = max(o2, x2);
                    if (v2 > f2) {
                        f2 = v2;
                        f2p = uint32_t(x - pb);
                        f2x = x2 > o2 ? 1u : 0u;
                        ft = (ft & ~8u) | (x2 == o2 ? 8u : 0u);
                    }
                }
                pw = (dp << 9) | (f1p << 16) | (f2p << 23);
                }
                if (dg > hn) {
                    hn = dg;
                    d = SRC_DIAG;
                }
                if (f1 > hn) {
                    hn = f1;
                    d = SRC_F1;
                }
                // (the two pieces score the same: the traceback takes the predecessor that comes first in the node's edge list,
                // whichever piece it offers; on a chain row both are predecessor 0 and the first piece stands)
                if (f2 > hn || (f2 == hn && d == SRC_F1 && f2p < f1p)) {
                    hn = f2;
                    d = SRC_F2;
                }
                d |= (f1x << 7) | (f2x << 8) | ft | pw;
            }
            POA_TICK(0)
            // ---- prefix maxima of Hn[x] - e x and Hn[x] - c x over the columns left of j ----
            // (the two scans stage by stage side by side: a wave issues in order, and every stage waits for the one before it)
            int sx = hn + ofs_e, sy = hn + ofs_c;  // (hn - e j and hn - c j; far below everything for the lanes past the last column)
            wave_prefix_max2(sx, sy);
            asm volatile("" : "+v"(sx), "+v"(sy));  // (keeps the last stage a v_max_i32_dpp: merged with the carry into a v_max3 it needs a move and a constant)
            POA_TICK(3)
            // ---- the carry of the wave on the left (or of the tile on the left): second half first, until it holds this row's tag ----
            int4 cin;
            {
                const uint32_t ca = carry_in + uint32_t(t) * 16u;
                for (;;) {
                    uint64_t hb, ha;
                    asm volatile("ds_read_b64 %0, %2 offset:8\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(hb), "=&v"(ha) : "v"(ca) : "memory");
                    cin = int4{int(uint32_t(ha)), int(uint32_t(ha >> 32)), int(uint32_t(hb >> 32)) & 3, int(uint32_t(hb))};  // (prefix maxima, flags, H: the order the code below grew up with)
                    if (__builtin_amdgcn_readfirstlane(int(uint32_t(hb >> 32)) >> 2) == t + 1) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const int vx = max(sx, cin.x);
            const int vy = max(sy, cin.y);
            const int ex = __builtin_amdgcn_update_dpp(cin.x, vx, 0x138, 0xf, 0xf, false);  // wave_shr:1: the lane on the left
            const int ey = __builtin_amdgcn_update_dpp(cin.y, vy, 0x138, 0xf, 0xf, false);
            // Is this column a STRICT new prefix maximum (per piece)?  The gap that ends one column to the right was opened here —
            // rather than extended — exactly then: a tie between extending and opening counts as extended (the traceback follows a
            // horizontal gap for as long as it can have been extended).  The flags travel right like the other carries.
            const int snm = ((hn + ofs_e) > ex ? 1 : 0) | ((hn + ofs_c) > ey ? 2 : 0);
            const int fl = __builtin_amdgcn_update_dpp(cin.z, snm, 0x138, 0xf, 0xf, false);
            int h = hn;
            uint32_t eb = 0;  // [2:0] E1 / E2 if a horizontal gap wins H, [3] E1 extended, [4] E2 extended
            if (j > 0) {
                // (ex / ey are real prefix maxima here: the one POA_NEG, the left edge of the matrix, meets column 0 only)
                const int e1 = ex + S.g + (j - 1) * S.e;
                const int e2 = ey + S.q + (j - 1) * S.c;
                const uint32_t e1x = (fl & 1) ? 0u : 1u;  // opened iff column j - 1 alone holds the maximum
                const uint32_t e2x = (fl & 2) ? 0u : 1u;
                if (e1 > h) {
                    h = e1;
                    eb = SRC_E1;
                }
                if (e2 > h) {
                    h = e2;
                    eb = SRC_E2;
                }
                eb |= (e1x << 3) | (e2x << 4);
            }
            POA_TICK(1)
            const int sl = r & (POA_RING - 1);
            sH[sl][tid] = h;
            sF1[sl][tid] = f1;
            sF2[sl][tid] = f2;
            if (lane == 63 && wave < POA_WAVES - 1) {  // first half, then the half with the tag (in this order: see s_carry)
                const uint64_t ha = uint64_t(uint32_t(vx)) | (uint64_t(uint32_t(vy)) << 32);
                const uint64_t hb = uint64_t(uint32_t(h)) | (uint64_t(uint32_t(snm) | (uint32_t(t + 1) << 2)) << 32);
                const uint32_t ca = carry_out + uint32_t(t) * 16u;
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(ca), "v"(ha), "v"(hb) : "memory");
            }
            if (active) {
                // (row bases are uniform: scalar arithmetic, the lane adds its 32-bit column offset)
                // (the row's base stays in scalar registers — the empty asm keeps the compiler from folding it into a per-lane 64-bit
                // pointer, which costs a 64-bit add or a multiply-add per store —, the lane adds a 32-bit offset: SADDR stores)
                const int64_t ro = int64_t(r) * W;
                const int ks = __builtin_amdgcn_readfirstlane(my_slot);
                // (the lane's 32-bit offsets are made opaque where they are used: hoisted out of the loop their zero extension is a
                // 64-bit pair in another block, and the store falls back to a 64-bit address per lane)
                if (ks >= 0) {  // some later row reads this one from memory (it is above that row's tile or out of its ring)
                    const int64_t ko = int64_t(ks) * W;
                    uint64_t bh = uint64_t(H + ko), b1 = uint64_t(F1 + ko), b2 = uint64_t(F2 + ko);
                    uint32_t oh = jb4, of1 = jb4, of2 = jb4;
                    asm volatile("" : "+s"(bh), "+s"(b1), "+s"(b2), "+v"(oh), "+v"(of1), "+v"(of2));
                    *(GI32*)((GU8*)bh + oh) = h;
                    *(GI32*)((GU8*)b1 + of1) = f1;
                    *(GI32*)((GU8*)b2 + of2) = f2;
                }
                uint64_t bd = uint64_t(dirs + ro), be = uint64_t(ebits + ro);
                uint32_t o4 = jb4, o1 = uint32_t(j);
                asm volatile("" : "+s"(bd), "+s"(be), "+v"(o4), "+v"(o1));
                *(GU32*)((GU8*)bd + o4) = d;
                *((GU8*)be + o1) = uint8_t(eb);
                if (h > my_best) {  // rows ascend in time: the first row wins ties (the column is the lane's own: set after the loop)
                    my_best = h;
                    my_r = r;
                }
                if (tid == POA_CB - 1 || j == W - 1) {
                    GI32* co = cout_row + int64_t(r) * 4;
                    co[0] = vx;
                    co[1] = vy;
                    co[2] = snm;
                    co[3] = h;
                }
            }
            edge_prev = cin.w;
            ph = h;
            p1 = f1;
            p2 = f2;
            pb = pe;
            pe = __builtin_amdgcn_readfirstlane(pe_nv);
            bs = bs_n;
            pr0 = pr0_n;
            my_slot = slot_n;
            POA_TICK(2)
        }
    }
#undef sH
#undef sF1
#undef sF2
#ifdef POA_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && diag == 10)
        printf("tile profile (wave %d, %d rows): predecessors %llu, scan + E %llu, stores %llu, waiting for the wave on the left %llu cycles\n", wave, nrows, tp[0], tp[1], tp[2], tp[3]);
#endif
    if (my_best > 0) my_j = j;
    // ---- the tile's best cell: maximum score, ties to the smallest row, then the smallest column ----
    auto better = [](int s1, int r1, int c1, int s2, int r2, int c2) {
        return s1 > s2 || (s1 == s2 && (r1 < r2 || (r1 == r2 && c1 < c2)));
    };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int s2 = __shfl_xor(my_best, o), r2 = __shfl_xor(my_r, o), c2 = __shfl_xor(my_j, o);
        if (better(s2, r2, c2, my_best, my_r, my_j)) {
            my_best = s2;
            my_r = r2;
            my_j = c2;
        }
    }
    if (lane == 0) {
        s_red[3 * wave] = my_best;
        s_red[3 * wave + 1] = my_r;
        s_red[3 * wave + 2] = my_j;
    }
    __syncthreads();
    if (tid == 0) {
        int bs = s_red[0], br = s_red[1], bc = s_red[2];
        for (int w2 = 1; w2 < POA_WAVES; ++w2)
            if (better(s_red[3 * w2], s_red[3 * w2 + 1], s_red[3 * w2 + 2], bs, br, bc)) {
                bs = s_red[3 * w2];
                br = s_red[3 * w2 + 1];
                bc = s_red[3 * w2 + 2];
            }
        GI32* tb = tile_best + (int64_t(cb) * nrb + rb) * 4;
        tb[0] = bs;
        tb[1] = br;
        tb[2] = bc;
        tb[3] = 0;
    }
}

// best cell of each job over its tiles: maximum score, ties to the smallest row, then the smallest column
__global__ void __launch_bounds__(256) k_poa_best(const PoaJob* __restrict__ jobs)
{
    __shared__ int4 s_b[256];
    const PoaJob J = jobs[blockIdx.x];
    const int nt = J.nrb * J.ncb;
    int4 b{0, 0, 0, 0};
    for (int t = threadIdx.x; t < nt; t += 256) {
        const int4 x = J.tile_best[t];
        if (x.x > b.x || (x.x == b.x && x.x > 0 && (x.y < b.y || (x.y == b.y && x.z < b.z)))) b = x;
    }
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (int(threadIdx.x) < o) {
            const int4 x = s_b[threadIdx.x + o], y = s_b[threadIdx.x];
            if (x.x > y.x || (x.x == y.x && x.x > 0 && (x.y < y.y || (x.y == y.y && x.z < y.z)))) s_b[threadIdx.x] = x;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        J.best[0] = s_b[0].x;
        J.best[1] = s_b[0].y;
        J.best[2] = s_b[0].z;
    }
}

// Walk back from the best cell.  out_node / out_pos receive the alignment in REVERSE order: (row, column - 1)
// for a diagonal move, (row, -1) for a node against a gap, (-1, column - 1) for a base against a gap;
// out_n = number of pairs.  One wave per alignment: every step is a chain of dependent loads (direction word,
// then the predecessor it names), so the 64 lanes read the cells (r - k, j - k) ahead of the walk and the run of
// plain diagonal steps through consecutive rows they confirm is emitted at once; whatever else comes next is
// one ordinary step.
__global__ void __launch_bounds__(64) k_poa_trace(const PoaJob* __restrict__ jobs)
{
    const PoaJob J = jobs[blockIdx.x];
    const int32_t* __restrict__ pred_off = J.pred_off;
    const int32_t* __restrict__ pred = J.pred;
    const uint32_t* __restrict__ dirs = J.dirs;
    const uint8_t* __restrict__ ebits = J.ebits;
    const int* __restrict__ best = J.best;
    int32_t* __restrict__ out_node = J.out_node;
    int32_t* __restrict__ out_pos = J.out_pos;
    const int cap = J.cap;
    const int64_t W = J.L + 1;
    const int lane = threadIdx.x;
    int r = best[1], j = best[2], n = 0;
    int state = 0;  // 0 H, 1 Hn (H without the horizontal sources), 2 F1, 3 F2, 4 E1, 5 E2
    bool entered = false;  // the vertical gap was entered from H by the step before: its extension / opening tie goes to the extension
    while (r > 0 && n < cap) {
        if (state == 0) {
            const int rr = r - lane, jl = j - lane;
            bool ok = rr > 0 && jl > 0;
            int pr = 0;
            if (ok) {
                const uint32_t d = dirs[int64_t(rr) * W + jl];
                const uint32_t eb = ebits[int64_t(rr) * W + jl];
                ok = (eb & 7u) == 0 && (d & 3u) == SRC_DIAG;
                if (ok) pr = pred[pred_off[rr] + int((d >> 9) & 127u)];
            }
            const uint64_t okm = __ballot(ok), contm = __ballot(ok && pr == rr - 1);
            const int fc = ~contm ? __builtin_ctzll(~contm) : 64;             // first lane that leaves the diagonal run
            int m = (fc < 64 && ((okm >> fc) & 1ull)) ? fc + 1 : fc;           // its own step is still a diagonal one
            m = min(m, cap - n);
            if (m > 0) {
                if (lane < m) {
                    out_node[n + lane] = rr;
                    out_pos[n + lane] = jl - 1;
                }
                r = __shfl(pr, m - 1);
                j -= m;
                n += m;
                continue;
            }
        }
        const uint32_t d = dirs[int64_t(r) * W + j];
        const uint32_t eb = ebits[int64_t(r) * W + j];
        const int pb = pred_off[r];
        if (state == 0 || state == 1) {
            const uint32_t src = (state == 0 && (eb & 7u)) ? eb & 7u : d & 3u;
            if (src == SRC_STOP) break;
            if (src == SRC_DIAG) {
                if (lane == 0) {
                    out_node[n] = r;
                    out_pos[n] = j - 1;
                }
                ++n;
                r = pred[pb + int((d >> 9) & 127u)];
                --j;
                state = 0;
            } else {
                state = int(src);  // F1 / F2 / E1 / E2: same cell, other matrix
                entered = true;
            }
        } else if (state == 2 || state == 3) {
            if (lane == 0) {
                out_node[n] = r;
                out_pos[n] = -1;
            }
            ++n;
            bool ext = state == 2 ? (d >> 7) & 1u : (d >> 8) & 1u;
            if (entered && (state == 2 ? (d >> 2) & 1u : (d >> 3) & 1u)) ext = true;
            entered = false;
            r = pred[pb + int(state == 2 ? (d >> 16) & 127u : (d >> 23) & 127u)];
            if (!ext) state = 0;
        } else {
            if (lane == 0) {
                out_node[n] = -1;
                out_pos[n] = j - 1;
            }
            ++n;
            const bool ext = state == 4 ? (eb >> 3) & 1u : (eb >> 4) & 1u;
            --j;
            if (!ext) state = 1;  // the gap was opened from Hn of the column to the left
        }
    }
    if (lane == 0) *J.out_n = n;
}

std::atomic<int64_t> g_row_stats[4];  // rows planned: chain rows, rows with 1-2 / more predecessors all in the LDS ring, rows that read memory
bool g_row_stats_on = false;

// ---- host side: the graph ----------------------------------------------------------------------------------
// Flat: every member is one array over the nodes, the edges or a pool, so that a copy of a graph (the snapshot a deferred
// consensus pass takes of every graph it is about to touch, and its release) is a dozen memcpys instead of three heap
// vectors per node — 3.4 s of a 31 250-read leaf's 16.9 s went into those copies.  Edge lists and aligned-node lists are
// singly linked in INSERTION order (the DP's and the consensus' tie-breaks follow that order).
struct PGraph {
    std::vector<char> base;
    std::vector<int> in_head, in_tail, out_head, out_tail, in_deg;  // per node: edge ids (-1: none), number of in-edges
    std::vector<int> al_head, al_tail;                              // per node: its aligned nodes (same position, another letter), pool entries
    std::vector<int> al_val, al_next;                               // the pool
    std::vector<int> e_from, e_to, e_in_next, e_out_next;           // per edge
    std::vector<int64_t> e_w;
    std::vector<int> rank;  // topological order: rank[i] = node id
    int nseq = 0;
    // a graph seeded with one sequence is a chain: it is built when somebody first looks at it (most clusters of a
    // batch stay singletons, and 16.7 k nodes with their edge lists are not free)
    std::string seed;
    int64_t seed_w = 0;
    bool seeded = false;
    size_t n_nodes() const { return base.size(); }
    size_t n_edges() const { return e_from.size(); }
    void seed_with(const char* s, int len, int64_t w)
    {
        if (len <= 0) return;
        seed.assign(s, size_t(len));
        seed_w = w;
        seeded = true;
        nseq = 1;
    }
    void ensure()
    {
        if (!seeded) return;
        seeded = false;
        nseq = 0;
        add_alignment({}, seed.data(), int(seed.size()), seed_w);
        std::string().swap(seed);
    }

    int add_node(char b)
    {
        base.push_back(b);
        in_head.push_back(-1);
        in_tail.push_back(-1);
        out_head.push_back(-1);
        out_tail.push_back(-1);
        in_deg.push_back(0);
        al_head.push_back(-1);
        al_tail.push_back(-1);
        return int(base.size()) - 1;
    }
    void al_push(int v, int x)  // x joins the aligned list of v
    {
        const int q = int(al_val.size());
        al_val.push_back(x);
        al_next.push_back(-1);
        if (al_tail[size_t(v)] >= 0)
            al_next[size_t(al_tail[size_t(v)])] = q;
        else
            al_head[size_t(v)] = q;
        al_tail[size_t(v)] = q;
    }
    void link_edge(int u, int v, int64_t w)  // a new edge at the end of u's out-list and v's in-list
    {
        const int e = int(e_from.size());
        e_from.push_back(u);
        e_to.push_back(v);
        e_w.push_back(w);
        e_in_next.push_back(-1);
        e_out_next.push_back(-1);
        if (out_tail[size_t(u)] >= 0)
            e_out_next[size_t(out_tail[size_t(u)])] = e;
        else
            out_head[size_t(u)] = e;
        out_tail[size_t(u)] = e;
        if (in_tail[size_t(v)] >= 0)
            e_in_next[size_t(in_tail[size_t(v)])] = e;
        else
            in_head[size_t(v)] = e;
        in_tail[size_t(v)] = e;
        in_deg[size_t(v)]++;
    }
    void add_edge(int u, int v, int64_t w)
    {
        for (int e = out_head[size_t(u)]; e >= 0; e = e_out_next[size_t(e)])
            if (e_to[size_t(e)] == v) {
                e_w[size_t(e)] += w;
                return;
            }
        link_edge(u, v, w);
    }
    // nodes for s[a..b), returns the first.  An edge between two consecutive bases of a sequence carries the weights of BOTH
    // (spoa: weights[i - 1] + weights[i]; every base of a sequence has the same weight here, src/consensus.cpp:15-32)
    int add_chain(const char* s, int a, int b, int64_t w)
    {
        int first = -1, prev = -1;
        for (int i = a; i < b; ++i) {
            const int v = add_node(s[i]);
            if (first < 0) first = v;
            if (prev >= 0) link_edge(prev, v, 2 * w);  // (both nodes are new: no such edge yet)
            prev = v;
        }
        return first;
    }
    // Topological order as spoa's graph keeps it (the order of the DP's rows decides which of several best cells is the first
    // one, and the consensus walks it): depth-first from every node in id order, a node after all its predecessors, the nodes
    // of one aligned column next to each other (the column is emitted when its first-reached member is complete).
    void toposort()
    {
        const size_t n = n_nodes();
        rank.clear();
        rank.reserve(n);
        std::vector<uint8_t> mark(n, 0), ignored(n, 0);
        std::vector<int> stack;
        for (size_t start = 0; start < n; ++start) {
            if (mark[start]) continue;
            stack.push_back(int(start));
            while (!stack.empty()) {
                const int cur = stack.back();
                bool valid = true;
                if (mark[size_t(cur)] != 2) {
                    for (int e = in_head[size_t(cur)]; e >= 0; e = e_in_next[size_t(e)]) {
                        const int t = e_from[size_t(e)];
                        if (mark[size_t(t)] != 2) {
                            stack.push_back(t);
                            valid = false;
                        }
                    }
                    if (!ignored[size_t(cur)])
                        for (int q = al_head[size_t(cur)]; q >= 0; q = al_next[size_t(q)]) {
                            const int a = al_val[size_t(q)];
                            if (mark[size_t(a)] != 2) {
                                stack.push_back(a);
                                ignored[size_t(a)] = 1;
                                valid = false;
                            }
                        }
                    if (valid) {
                        mark[size_t(cur)] = 2;
                        if (!ignored[size_t(cur)]) {
                            rank.push_back(cur);
                            for (int q = al_head[size_t(cur)]; q >= 0; q = al_next[size_t(q)]) rank.push_back(al_val[size_t(q)]);
                        }
                    } else {
                        mark[size_t(cur)] = 1;
                    }
                }
                if (valid) stack.pop_back();
            }
        }
    }
    // AddAlignment: aln = (node id or -1, position or -1) in forward order; an empty alignment adds a chain.  Node ids follow
    // spoa's creation order: the unaligned head of the read, its unaligned tail, then the aligned part base by base.
    void add_alignment(const std::vector<std::pair<int, int>>& aln, const char* s, int len, int64_t w)
    {
        if (len <= 0) return;
        planned = false;
        int first_pos = -1, last_pos = -1;
        for (auto& a : aln)
            if (a.second >= 0) {
                if (first_pos < 0) first_pos = a.second;
                last_pos = a.second;
            }
        if (first_pos < 0) {
            add_chain(s, 0, len, w);
            ++nseq;
            toposort();
            return;
        }
        int head = -1;
        if (first_pos > 0) {
            add_chain(s, 0, first_pos, w);
            head = int(n_nodes()) - 1;
        }
        const int tail = last_pos + 1 < len ? add_chain(s, last_pos + 1, len, w) : -1;
        std::vector<int> grp;
        for (auto& a : aln) {
            if (a.second < 0) continue;
            const char letter = s[a.second];
            int cur;
            if (a.first < 0) {
                cur = add_node(letter);
            } else if (base[size_t(a.first)] == letter) {
                cur = a.first;
            } else {
                cur = -1;
                for (int q = al_head[size_t(a.first)]; q >= 0; q = al_next[size_t(q)])
                    if (base[size_t(al_val[size_t(q)])] == letter) {
                        cur = al_val[size_t(q)];
                        break;
                    }
                if (cur < 0) {
                    cur = add_node(letter);
                    grp.clear();
                    for (int q = al_head[size_t(a.first)]; q >= 0; q = al_next[size_t(q)]) grp.push_back(al_val[size_t(q)]);
                    grp.push_back(a.first);
                    for (int x : grp) {
                        al_push(x, cur);
                        al_push(cur, x);
                    }
                }
            }
            if (head >= 0) add_edge(head, cur, 2 * w);
            head = cur;
        }
        if (tail >= 0 && head >= 0) add_edge(head, tail, 2 * w);
        ++nseq;
        toposort();
    }
    // rows = nodes in topological order (+1: row 0 is the virtual source); predecessor rows; which rows keep their
    // H / F1 / F2 in memory: those a later row cannot find in its tile's LDS ring (it sits in a lower tile or more
    // than POA_RING rows on).  On a chain that is one row in 64.  p_base / p_pslot: what the kernels read per row /
    // per predecessor entry besides (bases in row order, the plane row of each predecessor) — the batch layout copies
    // the five arrays as they are.
    std::vector<int32_t> p_poff, p_pred, p_slot, p_pslot;
    std::vector<uint8_t> p_base;
    int32_t p_nkeep = 1, p_max_preds = 0;
    bool planned = false;
    void plan()
    {
        const int R = int(n_nodes());
        std::vector<int> row_of(static_cast<size_t>(R), 0);
        for (int i = 0; i < R; ++i) row_of[size_t(rank[size_t(i)])] = i + 1;
        p_poff.assign(size_t(R) + 2, 0);
        p_pred.clear();
        p_base.assign(size_t(R) + 1, 0);
        p_max_preds = 0;
        std::vector<uint8_t> keep(size_t(R) + 1, 0);
        keep[0] = 1;
        for (int i = 0; i < R; ++i) {
            const int u = rank[size_t(i)];
            const int q = i + 1;
            p_base[size_t(q)] = uint8_t(base[size_t(u)]);
            p_max_preds = std::max(p_max_preds, in_deg[size_t(u)]);
            p_poff[size_t(q)] = int32_t(p_pred.size());
            if (in_head[size_t(u)] < 0) p_pred.push_back(0);
            for (int e = in_head[size_t(u)]; e >= 0; e = e_in_next[size_t(e)]) {
                const int pr = row_of[size_t(e_from[size_t(e)])];
                p_pred.push_back(pr);
                if ((pr - 1) / POA_RB != (q - 1) / POA_RB || q - pr > POA_RING) keep[size_t(pr)] = 1;
            }
        }
        p_poff[size_t(R) + 1] = int32_t(p_pred.size());
        if (g_row_stats_on) {  // (IOC_TRACE: what kinds of rows the tile kernel meets)
            int64_t k[4] = {0, 0, 0, 0};
            for (int q = 1; q <= R; ++q) {
                const int a = p_poff[size_t(q)], b = p_poff[size_t(q) + 1];
                bool ring = true;
                for (int x = a; x < b; ++x) {
                    const int pr = p_pred[size_t(x)];
                    if ((pr - 1) / POA_RB != (q - 1) / POA_RB || q - pr > POA_RING || pr == 0) ring = false;
                }
                if (b - a == 1 && ring && p_pred[size_t(a)] == q - 1)
                    k[0]++;
                else if (ring)
                    k[b - a <= 2 ? 1 : 2]++;
                else
                    k[3]++;
            }
            for (int x = 0; x < 4; ++x) g_row_stats[x].fetch_add(k[x], std::memory_order_relaxed);
        }
        p_slot.assign(size_t(R) + 1, -1);
        p_nkeep = 0;
        for (int r = 0; r <= R; ++r)
            if (keep[size_t(r)]) p_slot[size_t(r)] = p_nkeep++;
        p_pslot.resize(p_pred.size());
        for (size_t y = 0; y < p_pred.size(); ++y) p_pslot[y] = p_slot[size_t(p_pred[y])];
        planned = true;
    }
    // Heaviest bundle with branch completion (Lee 2003, as spoa's graph does it): in topological order every node takes its
    // heaviest in-edge — on equal weights the LATER edge wins unless its tail scores lower —, a node without in-edges scores -1;
    // the best-scoring node ends the bundle; if it is not a sink, the rest of the graph is scored again with the competitors of
    // its successors switched off, until a sink is reached.
    int branch_completion(size_t rk, std::vector<int64_t>& score, std::vector<int>& from) const
    {
        const int start = rank[rk];
        for (int e = out_head[size_t(start)]; e >= 0; e = e_out_next[size_t(e)])
            for (int f = in_head[size_t(e_to[size_t(e)])]; f >= 0; f = e_in_next[size_t(f)])
                if (e_from[size_t(f)] != start) score[size_t(e_from[size_t(f)])] = -1;
        int best = -1;
        for (size_t i = rk + 1; i < rank.size(); ++i) {
            const int u = rank[i];
            score[size_t(u)] = -1;
            from[size_t(u)] = -1;
            for (int e = in_head[size_t(u)]; e >= 0; e = e_in_next[size_t(e)]) {
                const int f = e_from[size_t(e)];
                if (score[size_t(f)] == -1) continue;
                const int64_t w = e_w[size_t(e)];
                if (score[size_t(u)] < w || (score[size_t(u)] == w && score[size_t(from[size_t(u)])] <= score[size_t(f)])) {
                    score[size_t(u)] = w;
                    from[size_t(u)] = f;
                }
            }
            if (from[size_t(u)] >= 0) score[size_t(u)] += score[size_t(from[size_t(u)])];
            if (best < 0 || score[size_t(best)] < score[size_t(u)]) best = u;
        }
        return best;
    }
    std::string consensus() const
    {
        const size_t n = n_nodes();
        if (n == 0 || rank.empty()) return std::string();
        std::vector<int64_t> score(n, -1);
        std::vector<int> from(n, -1);
        int best = -1;
        for (int u : rank) {
            for (int e = in_head[size_t(u)]; e >= 0; e = e_in_next[size_t(e)]) {
                const int f = e_from[size_t(e)];
                const int64_t w = e_w[size_t(e)];
                if (score[size_t(u)] < w || (score[size_t(u)] == w && score[size_t(from[size_t(u)])] <= score[size_t(f)])) {
                    score[size_t(u)] = w;
                    from[size_t(u)] = f;
                }
            }
            if (from[size_t(u)] >= 0) score[size_t(u)] += score[size_t(from[size_t(u)])];
            if (best < 0 || score[size_t(best)] < score[size_t(u)]) best = u;
        }
        if (out_head[size_t(best)] >= 0) {
            std::vector<size_t> row_of(n, 0);
            for (size_t i = 0; i < rank.size(); ++i) row_of[size_t(rank[i])] = i;
            while (out_head[size_t(best)] >= 0) best = branch_completion(row_of[size_t(best)], score, from);
        }
        std::string s;
        for (int u = best; u >= 0; u = from[size_t(u)]) s.push_back(base[size_t(u)]);
        std::reverse(s.begin(), s.end());
        return s;
    }
};

}  // namespace

struct PoaPending {
    std::string seq;
    int64_t weight = 1;
    int tag = -1;         // the driver's entry that caused the operation (ioc_consensus_spec_ops), -1: untagged
    bool marker = false;  // not an addition: "take the consensus here" (consensus_deferred)
};

struct ioc_poa {
    ioc_ctx* ctx = nullptr;
    PoaScores S{4, -8, -8, -4, -20, -1};
    int pred_lds = POA_PRED_LDS;  // IOC_POA_PRED_LDS: a smaller staging area, for the tests of the overflow path
    std::map<int, PGraph> g[2];
    // additions not aligned yet: graphs are independent until somebody reads one, so additions are queued and the
    // queues of all graphs are worked off together, one addition per graph and round, in batched launches
    std::map<int, std::vector<PoaPending>> pending[2];
    bool lazy = true;
    // deferred consensus (ioc_consensus_spec_ops): results by (side, idx, tag); what a graph and its queue were before the
    // first speculative operation was APPLIED to it (graphs that did not exist: existed = false)
    std::map<std::tuple<int, int, int>, std::string> deferred;
    struct Snap {
        bool existed = false;
        int max_tag = -1;  // the latest entry whose operation was applied to the graph since the snapshot
        PGraph g;
        std::vector<PoaPending> q;
        // a graph (re)created by a tagged operation of the pass: the seed and every operation issued to it since are kept,
        // so that a rollback to a point AFTER its creation can rebuild it (create_tag < first_tag)
        int create_tag = -1;
        std::string create_seq;
    };
    std::map<int, Snap> snap[2];
    DevBuf d_int, d_dirs, d_eb, d_carry, d_tbest, d_small, d_aln, d_jobs;
    // the alignment of the last addition worked off (tests / inspection): node ids, positions, score
    std::vector<int32_t> last_node, last_pos;
    int32_t last_score = 0;
    int64_t n_batches = 0, n_aligned = 0;
    double ms_layout = 0, ms_alloc = 0, ms_gpu = 0, ms_graph = 0;  // IOC_TRACE: host layout of the batches, launches + copies, AddAlignment
    double ms_snap = 0, ms_mark = 0, ms_plan = 0, ms_post = 0;     // snapshots, consensus at markers, row plans, alignments back into node ids
    double ms_dev[3] = {0, 0, 0};                                  // (events) uploads, kernels, downloads
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::string err;
    size_t reserve_hint = 0;  // the memory budget of a batch (poa_flush): what a large buffer is sized for at once
};

namespace {

int poa_reserve(ioc_poa* p, DevBuf& b, size_t bytes, size_t hint = 0)
{
    if (b.cap >= bytes) return IOC_OK;
    const size_t old_cap = b.cap;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    // hipMalloc costs ~0.1 ms per MB (3.4 s of a 31 250-read batch's 17 s went there while buffers grew by factors towards a
    // 48 GB budget).  Batches grow as the graphs do and — with deferred consensus — as the passes do, up to the budget of
    // poa_flush: small buffers double; past a gigabyte the next size is `hint`, the buffer's share of that budget (one more
    // allocation, the last), or half as much again.
    size_t want = bytes + bytes / 8 + 4096;
    if (want < (size_t(1) << 30))
        want = std::max(want, std::min(2 * old_cap, size_t(1) << 30));
    else
        want = std::max(want, hint ? hint : old_cap + old_cap / 2);
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        want = bytes + 4096;
        b.p = nullptr;
    }
    if (!b.p && hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        (void)hipGetLastError();
        return ioc_fail(p->ctx, IOC_ERR_CAPACITY, "POA: hipMalloc of the DP matrices failed (" + std::to_string(want >> 20) + " MB)");
    }
    b.cap = want;
    return IOC_OK;
}

#define PCHK(p, call)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return ioc_fail((p)->ctx, IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct HostJob {
    PGraph* G = nullptr;
    const PoaPending* item = nullptr;
    std::vector<std::pair<int, int>> aln;  // out: forward order, NODE IDS
    int32_t score = 0;
    std::string cons;  // the consensus right after this addition, when the queue asks for it next (computed with the graph update, in parallel)
    bool have_cons = false;
    size_t bytes() const  // device memory of the alignment: direction word + E byte per cell, 3 planes of kept rows
    {
        const size_t W = item->seq.size() + 1, R1 = G->n_nodes() + 1;
        return R1 * W * 5 + size_t(G->p_nkeep) * W * 12 + (W / POA_CB + 1) * R1 * sizeof(int4) + (size_t(1) << 20);
    }
};

// align every job's read to its graph: one batch of launches
int poa_align_batch(ioc_poa* p, std::vector<HostJob>& jobs)
{
    ioc_ctx* c = p->ctx;
    PCHK(p, hipSetDevice(c->device));
    const size_t K = jobs.size();
    if (K == 0) return IOC_OK;
    hipStream_t s = c->stream;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now();
    // ---- layout of the batch in the arenas ----
    struct Lay {
        size_t cells, o_cells, o_plane, o_pslot, o_slot, o_carry, o_tbest, o_small, o_base, o_poff, o_pred, o_seq, o_best, o_n, o_aln;
        int R, L, nrb, ncb, cap;
    };
    std::vector<Lay> lay(K);
    std::vector<uint8_t> small;
    size_t tot_cells = 0, tot_plane = 0, tot_carry = 0, tot_tbest = 0, tot_aln = 0;
    int max_w = 1, max_diag = 0, max_ncb = 1;
    for (size_t x = 0; x < K; ++x) {
        const PGraph& G = *jobs[x].G;
        const std::string& seq = jobs[x].item->seq;
        Lay& l = lay[x];
        l.R = int(G.n_nodes());
        l.L = int(seq.size());
        if (l.L + 1 > POA_MAX_COLS) return ioc_fail(c, IOC_ERR_CAPACITY, "POA: sequences above 2^20 bases are not supported");
        l.ncb = (l.L + 1 + POA_CB - 1) / POA_CB;
        l.nrb = (l.R + POA_RB - 1) / POA_RB;
        l.cap = l.R + l.L + 2;
        l.cells = size_t(l.R + 1) * size_t(l.L + 1);
        l.o_cells = tot_cells;
        tot_cells += (l.cells + 63) & ~size_t(63);
        l.o_plane = tot_plane;
        tot_plane += (size_t(jobs[x].G->p_nkeep) * size_t(l.L + 1) + 63) & ~size_t(63);
        l.o_carry = tot_carry;
        tot_carry += size_t(l.ncb) * size_t(l.R + 1);
        l.o_tbest = tot_tbest;
        tot_tbest += size_t(l.ncb) * size_t(l.nrb);
        l.o_aln = tot_aln;
        tot_aln += size_t(l.cap) * 2;
        max_w = std::max(max_w, l.L + 1);
        max_diag = std::max(max_diag, l.nrb + l.ncb - 1);
        max_ncb = std::max(max_ncb, l.ncb);
        // small arrays: bases in topological order, predecessor rows and their plane rows, the read
        auto align16 = [&]() { small.resize((small.size() + 15) & ~size_t(15)); };
        auto put32 = [&](const std::vector<int32_t>& v) {
            align16();
            const size_t o = small.size();
            small.insert(small.end(), reinterpret_cast<const uint8_t*>(v.data()), reinterpret_cast<const uint8_t*>(v.data() + v.size()));
            return o;
        };
        if (G.p_max_preds > POA_MAX_PREDS) return ioc_fail(c, IOC_ERR_CAPACITY, "POA: a node with more than 127 predecessors");
        auto put8 = [&](const std::vector<uint8_t>& v) {
            align16();
            const size_t o = small.size();
            small.insert(small.end(), v.begin(), v.end());
            return o;
        };
        l.o_base = put8(G.p_base);
        l.o_poff = put32(G.p_poff);
        l.o_pred = put32(G.p_pred);
        l.o_pslot = put32(G.p_pslot);
        l.o_slot = put32(G.p_slot);
        l.o_seq = small.size();
        small.insert(small.end(), seq.begin(), seq.end());
        align16();
        l.o_best = small.size();
        small.resize(small.size() + 16, 0);
        l.o_n = small.size();
        small.resize(small.size() + 16, 0);
    }
    int r;
    const double t_res = now();
    if ((r = poa_reserve(p, p->d_int, tot_plane * 12)) != IOC_OK) return r;   // H, F1, F2 planes of the kept rows
    if ((r = poa_reserve(p, p->d_dirs, tot_cells * 4, p->reserve_hint / 5 * 4)) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_eb, tot_cells, p->reserve_hint / 5)) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_carry, tot_carry * sizeof(int4))) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_tbest, tot_tbest * sizeof(int4))) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_small, small.size())) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_aln, tot_aln * 4)) != IOC_OK) return r;
    if ((r = poa_reserve(p, p->d_jobs, K * sizeof(PoaJob))) != IOC_OK) return r;
    p->ms_alloc += now() - t_res;
    uint8_t* sm = static_cast<uint8_t*>(p->d_small.p);
    int* ints = static_cast<int*>(p->d_int.p);
    std::vector<PoaJob> dj(K);
    for (size_t x = 0; x < K; ++x) {
        const Lay& l = lay[x];
        PoaJob& j = dj[x];
        j.R = l.R;
        j.L = l.L;
        j.nrb = l.nrb;
        j.ncb = l.ncb;
        j.base = sm + l.o_base;
        j.pred_off = reinterpret_cast<const int32_t*>(sm + l.o_poff);
        j.pred = reinterpret_cast<const int32_t*>(sm + l.o_pred);
        j.pred_slot = reinterpret_cast<const int32_t*>(sm + l.o_pslot);
        j.slot = reinterpret_cast<const int32_t*>(sm + l.o_slot);
        j.seq = sm + l.o_seq;
        j.H = ints + l.o_plane;
        j.F1 = ints + tot_plane + l.o_plane;
        j.F2 = ints + 2 * tot_plane + l.o_plane;
        j.dirs = static_cast<uint32_t*>(p->d_dirs.p) + l.o_cells;
        j.ebits = static_cast<uint8_t*>(p->d_eb.p) + l.o_cells;
        j.carry = static_cast<int4*>(p->d_carry.p) + l.o_carry;
        j.tile_best = static_cast<int4*>(p->d_tbest.p) + l.o_tbest;
        j.best = reinterpret_cast<int*>(sm + l.o_best);
        j.out_node = static_cast<int32_t*>(p->d_aln.p) + l.o_aln;
        j.out_pos = j.out_node + l.cap;
        j.out_n = reinterpret_cast<int32_t*>(sm + l.o_n);
        j.cap = l.cap;
    }
    const double t_gpu = now();
    p->ms_layout += t_gpu - t_in;
    const bool trace = getenv("IOC_TRACE") != nullptr;
    if (trace && !p->ev[0])
        for (auto& e : p->ev) PCHK(p, hipEventCreate(&e));
    if (trace) PCHK(p, hipEventRecord(p->ev[0], s));
    PCHK(p, hipMemcpyAsync(sm, small.data(), small.size(), hipMemcpyHostToDevice, s));
    PCHK(p, hipMemcpyAsync(p->d_jobs.p, dj.data(), K * sizeof(PoaJob), hipMemcpyHostToDevice, s));
    if (trace) PCHK(p, hipEventRecord(p->ev[1], s));
    const PoaJob* djobs = static_cast<const PoaJob*>(p->d_jobs.p);
    hipLaunchKernelGGL(k_poa_init, dim3(unsigned((max_w + 255) / 256), unsigned(K)), dim3(256), 0, s, djobs);
    PCHK(p, hipGetLastError());
    for (int dg = 0; dg < max_diag; ++dg) {  // one anti-diagonal of tiles (of every job) per launch
        hipLaunchKernelGGL(k_poa_tile, dim3(unsigned(std::min(max_ncb, dg + 1)), unsigned(K)), dim3(POA_THREADS), 0, s, djobs, dg, p->S, p->pred_lds);
        PCHK(p, hipGetLastError());
    }
    hipLaunchKernelGGL(k_poa_best, dim3(unsigned(K)), dim3(256), 0, s, djobs);
    PCHK(p, hipGetLastError());
    hipLaunchKernelGGL(k_poa_trace, dim3(unsigned(K)), dim3(64), 0, s, djobs);
    PCHK(p, hipGetLastError());
    if (trace) PCHK(p, hipEventRecord(p->ev[2], s));
    std::vector<uint8_t> back(small.size());
    std::vector<int32_t> haln(tot_aln);
    PCHK(p, hipMemcpyAsync(back.data(), sm, small.size(), hipMemcpyDeviceToHost, s));
    PCHK(p, hipMemcpyAsync(haln.data(), p->d_aln.p, tot_aln * 4, hipMemcpyDeviceToHost, s));
    if (trace) PCHK(p, hipEventRecord(p->ev[3], s));
    PCHK(p, hipStreamSynchronize(s));
    p->ms_gpu += now() - t_gpu;
    if (trace)
        for (int x = 0; x < 3; ++x) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p->ev[x], p->ev[x + 1]) == hipSuccess) p->ms_dev[x] += double(ms);
        }
    const double t_post = now();
    ioc_parallel_for(K, [&](size_t x) {
        const Lay& l = lay[x];
        const PGraph& G = *jobs[x].G;
        int32_t hb[3], hn;
        memcpy(hb, back.data() + l.o_best, 12);
        memcpy(&hn, back.data() + l.o_n, 4);
        jobs[x].score = hb[0];
        jobs[x].aln.clear();
        jobs[x].aln.reserve(size_t(hn));
        const int32_t* nd = haln.data() + l.o_aln;
        const int32_t* ps = nd + l.cap;
        for (int i = hn - 1; i >= 0; --i) jobs[x].aln.emplace_back(nd[i] > 0 ? G.rank[size_t(nd[i]) - 1] : -1, ps[i]);
    }, 16);
    p->ms_post += now() - t_post;
    p->n_batches++;
    p->n_aligned += int64_t(K);
    return IOC_OK;
}

// work off every queue: one addition per graph and round, as many graphs per batch as the memory budget allows
// a consensus marker at the head of a queue: the consensus of the graph as it is now
void poa_take_markers(ioc_poa* p, int side, int idx, std::vector<PoaPending>& q, const std::string* ready = nullptr)
{
    const auto tm0 = std::chrono::steady_clock::now();
    const bool timed = !q.empty() && q.front().marker;
    while (!q.empty() && q.front().marker) {
        auto sn = p->snap[side].find(idx);
        if (sn != p->snap[side].end()) sn->second.max_tag = std::max(sn->second.max_tag, q.front().tag);
        auto it = p->g[side].find(idx);
        if (it != p->g[side].end()) {
            it->second.ensure();
            p->deferred[std::make_tuple(side, idx, q.front().tag)] = ready ? *ready : it->second.consensus();
        }
        q.erase(q.begin());
    }
    if (timed) p->ms_mark += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count();
}
void poa_snapshot(ioc_poa* p, int side, int idx)
{
    if (p->snap[side].count(idx)) return;
    const auto ts0 = std::chrono::steady_clock::now();
    ioc_poa::Snap s;
    auto it = p->g[side].find(idx);
    s.existed = it != p->g[side].end();
    if (s.existed) s.g = it->second;
    auto pq = p->pending[side].find(idx);
    if (pq != p->pending[side].end()) s.q = pq->second;
    p->snap[side][idx] = std::move(s);
    p->ms_snap += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count();
}

// only_marked: work off the queues of the graphs that hold a consensus marker (those graphs are snapshotted first) — as many
// rounds as they need; the head of any other queue rides along in those rounds while it is tagged below safe_tag (such an
// operation is never rolled back), so that plain additions do not pile up behind a graph's first consensus
int poa_flush(ioc_poa* p, bool only_marked = false, int safe_tag = INT32_MIN)
{
    auto marked = [&](const std::vector<PoaPending>& q) {
        for (auto& it : q)
            if (it.marker) return true;
        return false;
    };
    auto wanted = [&](const std::vector<PoaPending>& q) {
        if (q.empty()) return false;
        if (!only_marked) return true;
        return marked(q) || (!q.front().marker && q.front().tag < safe_tag);
    };
    bool any = false;
    for (int side = 0; side < 2; ++side)
        for (auto& kv : p->pending[side])
            if (!kv.second.empty() && (!only_marked || marked(kv.second))) {
                any = true;
                if (only_marked) poa_snapshot(p, side, kv.first);
            }
    if (!any) return IOC_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = size_t(8) << 30;
    for (;;) {
        const size_t have = p->d_int.cap + p->d_dirs.cap + p->d_eb.cap;
        std::vector<HostJob> jobs, cand;
        std::vector<std::pair<int, int>> who, who_c;
        size_t used = 0;
        if (only_marked) {  // rounds go on only while a graph with a consensus request has work left
            bool more = false;
            for (int side = 0; side < 2 && !more; ++side)
                for (auto& kv : p->pending[side])
                    if (marked(kv.second)) {
                        more = true;
                        break;
                    }
            if (!more) return IOC_OK;
        }
        for (int side = 0; side < 2; ++side)
            for (auto& kv : p->pending[side]) {
                if (!wanted(kv.second)) continue;
                poa_take_markers(p, side, kv.first, kv.second);
                if (kv.second.empty()) continue;
                if (only_marked && !marked(kv.second) && !(kv.second.front().tag < safe_tag)) continue;
                auto it = p->g[side].find(kv.first);
                if (it == p->g[side].end()) return ioc_fail(p->ctx, IOC_ERR_STATE, "POA: addition to a graph that does not exist");
                HostJob j;
                j.G = &it->second;
                j.item = &kv.second.front();
                cand.push_back(j);
                who_c.emplace_back(side, kv.first);
            }
        // seeded graphs are built, and the row plans of graphs that changed are renewed, on the host's cores
        const auto tp0 = std::chrono::steady_clock::now();
        ioc_parallel_for(cand.size(), [&](size_t x) {
            PGraph& G = *cand[x].G;
            G.ensure();
            if (!G.planned && G.n_nodes()) G.plan();
        }, 48);
        p->ms_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
        // The memory budget of a batch: 10 GB (≈ 400 additions of a 2 kb read to a 2.5 k-node graph, more than the chip runs
        // at a time; freeing and re-allocating tens of GB costs seconds and serialises concurrent processes in the driver),
        // or what 32 alignments of the round's largest take (a 16.7 kb read against its graph: 1.5 GB — with 10 GB, six to
        // a batch, such a batch took twice as long), within half of the free memory and 48 GB.
        size_t largest = 0;
        for (const HostJob& j : cand)
            if (j.G->n_nodes() && !j.item->seq.empty()) largest = std::max(largest, j.bytes());
        // ... but device memory is not free: hipMalloc costs ~0.07 ms per MB (a `cluster` process that merges two leaves — 1 600
        // alignments — spent 0.70 of its 2.4 s allocating the 10 GB), while a batch of a quarter the size only runs less efficiently
        // (bound by the rows' dependent chain instead of the VALU).  So the 10 GB are for a flush with enough work to pay for them:
        // a 24th of what is queued, between 2 and 10 GB, and never less than what the engine holds already.
        size_t queued = 0;
        for (int side = 0; side < 2; ++side)
            for (auto& kv : p->pending[side]) {
                auto it = p->g[side].find(kv.first);
                if (it == p->g[side].end()) continue;
                const size_t rows = it->second.n_nodes() + 1;
                for (auto& item : kv.second)
                    if (!item.marker) queued += rows * (item.seq.size() + 1) * 5;
            }
        const size_t base = std::max(have, std::min(size_t(10) << 30, std::max(size_t(2) << 30, queued / 24)));
        size_t budget = std::min(std::min((free_b + have) / 2, size_t(48) << 30), std::max(base, 32 * largest));
        if (const char* e = getenv("IOC_POA_BUDGET_MB")) budget = size_t(atoll(e)) << 20;
        p->reserve_hint = budget + budget / 16;  // (the cells' direction words are 4/5 of it, their E bytes 1/5)
        for (size_t x = 0; x < cand.size(); ++x) {
            HostJob& j = cand[x];
            if (!j.G->n_nodes() || j.item->seq.empty()) {  // nothing to align: a chain of its own
                jobs.push_back(j);
                who.push_back(who_c[x]);
                continue;
            }
            const size_t need = j.bytes();
            if (!jobs.empty() && (used + need > budget || jobs.size() >= 2048)) continue;  // next round
            used += need;
            jobs.push_back(j);
            who.push_back(who_c[x]);
        }
        if (jobs.empty()) return IOC_OK;
        std::vector<HostJob> run;
        std::vector<size_t> run_ix;
        for (size_t x = 0; x < jobs.size(); ++x)
            if (jobs[x].G->n_nodes() && !jobs[x].item->seq.empty()) {
                run.push_back(jobs[x]);
                run_ix.push_back(x);
            }
        int r = poa_align_batch(p, run);
        if (r != IOC_OK) return r;
        for (size_t y = 0; y < run.size(); ++y) {
            jobs[run_ix[y]].aln = std::move(run[y].aln);
            jobs[run_ix[y]].score = run[y].score;
        }
        const auto tg0 = std::chrono::steady_clock::now();
        // (the graphs of a round are different graphs: their updates — AddAlignment + topological sort — run on
        // the host's cores)
        ioc_parallel_for(jobs.size(), [&](size_t x) {
            HostJob& j = jobs[x];
            j.G->add_alignment(j.aln, j.item->seq.data(), int(j.item->seq.size()), j.item->weight);
            const auto& q = p->pending[who[x].first].find(who[x].second)->second;  // (j.item is its head)
            if (q.size() > 1 && q[1].marker) {
                j.cons = j.G->consensus();
                j.have_cons = true;
            }
        }, 16);
        for (size_t x = 0; x < jobs.size(); ++x) {
            HostJob& j = jobs[x];
            if (x + 1 == jobs.size()) {  // what ioc_poa_last_alignment reports
                p->last_node.clear();
                p->last_pos.clear();
                for (auto& a : j.aln) {
                    p->last_node.push_back(a.first);
                    p->last_pos.push_back(a.second);
                }
                p->last_score = j.score;
            }
            auto& q = p->pending[who[x].first][who[x].second];
            {
                auto sn = p->snap[who[x].first].find(who[x].second);
                if (sn != p->snap[who[x].first].end()) sn->second.max_tag = std::max(sn->second.max_tag, q.front().tag);
            }
            q.erase(q.begin());
            poa_take_markers(p, who[x].first, who[x].second, q, j.have_cons ? &j.cons : nullptr);
        }
        p->ms_graph += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count();
    }
}

int op_create(void* u, int side, int idx, const char* seq, int len)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    PGraph G;
    G.seed_with(seq, len, 1);
    p->g[side][idx] = std::move(G);
    p->pending[side].erase(idx);
    return 0;
}
int op_size(void* u, int side, int idx)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1) return -1;
    auto it = p->g[side].find(idx);
    if (it == p->g[side].end()) return -1;
    auto pq = p->pending[side].find(idx);
    int queued = 0;  // queued additions count (consensus markers do not)
    if (pq != p->pending[side].end())
        for (auto& x : pq->second) queued += !x.marker;
    return it->second.nseq + queued;
}
int op_add(void* u, int side, int idx, const char* seq, int len, unsigned weight)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    if (p->g[side].find(idx) == p->g[side].end()) return -1;
    if (len == 0) return 0;  // AddAlignment ignores an empty sequence
    PoaPending it;
    it.seq.assign(seq, size_t(len));
    it.weight = int64_t(weight);
    p->pending[side][idx].push_back(std::move(it));
    if (!p->lazy && poa_flush(p) != IOC_OK) return -1;
    return 0;
}
int op_consensus(void* u, int side, int idx, char* out, int cap)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1) return -1;
    if (poa_flush(p) != IOC_OK) return -1;  // this graph must be up to date; everybody else's queue rides along
    auto it = p->g[side].find(idx);
    if (it == p->g[side].end()) return -1;
    it->second.ensure();
    const std::string s = it->second.consensus();
    if (int(s.size()) > cap) return -1;
    memcpy(out, s.data(), s.size());
    return int(s.size());
}
int op_purge(void* u, int side, int idx, const char* seq, int len, unsigned weight)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1) return -1;
    PGraph G;
    G.seed_with(seq, len, int64_t(weight));  // ConsPurge: the representative alone, with the old count as weight
    p->g[side][idx] = std::move(G);
    p->pending[side].erase(idx);
    return 0;
}

// ---- ioc_consensus_spec_ops ------------------------------------------------------------------------------------
int sp_create(void* u, int side, int idx, const char* seq, int len, int tag)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1) return -1;
    poa_snapshot(p, side, idx);  // (a graph that did not exist: rollback erases it)
    ioc_poa::Snap& sn = p->snap[side][idx];
    sn.max_tag = std::max(sn.max_tag, tag);
    sn.create_tag = tag;
    sn.create_seq.assign(seq, size_t(len));
    sn.q.clear();
    return op_create(u, side, idx, seq, len);
}
int sp_add(void* u, int side, int idx, const char* seq, int len, unsigned weight, int tag)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    if (p->g[side].find(idx) == p->g[side].end()) return -1;
    if (len == 0) return 0;
    PoaPending it;
    it.seq.assign(seq, size_t(len));
    it.weight = int64_t(weight);
    it.tag = tag;
    {
        auto sn = p->snap[side].find(idx);
        if (sn != p->snap[side].end() && sn->second.create_tag >= 0) sn->second.q.push_back(it);  // (the log of a graph created in this pass)
    }
    p->pending[side][idx].push_back(std::move(it));
    return 0;
}
int sp_consensus_deferred(void* u, int side, int idx, int tag)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    if (side < 0 || side > 1 || p->g[side].find(idx) == p->g[side].end()) return -1;
    PoaPending it;
    it.marker = true;
    it.tag = tag;
    {
        auto sn = p->snap[side].find(idx);
        if (sn != p->snap[side].end() && sn->second.create_tag >= 0) sn->second.q.push_back(it);
    }
    p->pending[side][idx].push_back(std::move(it));
    return 0;
}
int sp_flush(void* u, int safe_tag) { return poa_flush(static_cast<ioc_poa*>(u), true, safe_tag) == IOC_OK ? 0 : -1; }
int sp_collect(void* u, int side, int idx, int tag, char* out, int cap)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    auto it = p->deferred.find(std::make_tuple(side, idx, tag));
    if (it == p->deferred.end()) {
        if (poa_flush(p, true) != IOC_OK) return -1;
        it = p->deferred.find(std::make_tuple(side, idx, tag));
        if (it == p->deferred.end()) return -1;
    }
    if (int(it->second.size()) > cap) return -1;
    memcpy(out, it->second.data(), it->second.size());
    return int(it->second.size());
}
int sp_rollback(void* u, int first_tag)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    for (int side = 0; side < 2; ++side) {
        for (auto& kv : p->snap[side]) {
            if (kv.second.max_tag < first_tag) continue;  // everything applied to it stands (e.g. a finalized event + purge)
            if (kv.second.create_tag >= 0 && kv.second.create_tag < first_tag) {
                // created earlier in the pass by an entry that stands: the seed again, with the operations issued since
                PGraph G;
                G.seed_with(kv.second.create_seq.data(), int(kv.second.create_seq.size()), 1);
                p->g[side][kv.first] = std::move(G);
                p->pending[side][kv.first] = std::move(kv.second.q);
                continue;
            }
            if (kv.second.existed)
                p->g[side][kv.first] = std::move(kv.second.g);
            else
                p->g[side].erase(kv.first);
            if (kv.second.q.empty())
                p->pending[side].erase(kv.first);
            else
                p->pending[side][kv.first] = std::move(kv.second.q);
        }
        p->snap[side].clear();
        // what was only queued: drop the operations of the entries that are walked again
        for (auto& kv : p->pending[side]) {
            auto& q = kv.second;
            q.erase(std::remove_if(q.begin(), q.end(), [&](const PoaPending& x) { return x.tag >= first_tag; }), q.end());
        }
    }
    for (auto it = p->deferred.begin(); it != p->deferred.end();)
        it = std::get<2>(it->first) >= first_tag ? p->deferred.erase(it) : std::next(it);
    return 0;
}
int sp_commit(void* u)
{
    ioc_poa* p = static_cast<ioc_poa*>(u);
    p->snap[0].clear();
    p->snap[1].clear();
    p->deferred.clear();
    return 0;
}
const ioc_consensus_spec_ops g_spec_ops = {sp_create, sp_add, sp_consensus_deferred, sp_flush, sp_collect, sp_rollback, sp_commit};

}  // namespace

extern "C" {

int ioc_poa_create(ioc_ctx* ctx, int32_t m, int32_t n, int32_t g, int32_t e, int32_t q, int32_t c, ioc_poa** out)
{
    if (!ctx || !out) return IOC_ERR_ARG;
    // the row scan computes horizontal gaps from H-without-gaps: valid when opening is no cheaper than extending
    if (!(g <= e && e <= 0 && q <= c && c <= 0 && n <= 0 && m >= 0))
        return ioc_fail(ctx, IOC_ERR_ARG, "POA scores: need m >= 0, n <= 0, g <= e <= 0, q <= c <= 0");
    ioc_poa* p = new ioc_poa;
    p->ctx = ctx;
    p->S = PoaScores{m, n, g, e, q, c};
    g_row_stats_on = getenv("IOC_TRACE") != nullptr;
    if (const char* e2 = getenv("IOC_POA_PRED_LDS")) p->pred_lds = std::max(1, std::min(POA_PRED_LDS, atoi(e2)));
    *out = p;
    return IOC_OK;
}

void ioc_poa_destroy(ioc_poa* p)
{
    if (!p) return;
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc] POA: %lld alignments in %lld batches; batch layout %.1f ms (%.1f ms of it device allocations), launches + copies %.1f ms, graph updates %.1f ms\n",
                (long long)p->n_aligned, (long long)p->n_batches, p->ms_layout, p->ms_alloc, p->ms_gpu, p->ms_graph);
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc] POA: device uploads %.1f ms, kernels %.1f ms, downloads %.1f ms; snapshots %.1f ms, consensus at markers %.1f ms, row plans %.1f ms, "
                        "alignments into node ids %.1f ms\n",
                p->ms_dev[0], p->ms_dev[1], p->ms_dev[2], p->ms_snap, p->ms_mark, p->ms_plan, p->ms_post);
    if (getenv("IOC_TRACE"))
        fprintf(stderr, "[ioc] POA: rows planned: %lld chain rows, %lld with 1-2 predecessors in the LDS ring, %lld with more, %lld that read memory\n",
                (long long)g_row_stats[0].load(), (long long)g_row_stats[1].load(), (long long)g_row_stats[2].load(), (long long)g_row_stats[3].load());
    for (auto& e : p->ev)
        if (e) (void)hipEventDestroy(e);
    for (DevBuf* b : {&p->d_int, &p->d_dirs, &p->d_eb, &p->d_carry, &p->d_tbest, &p->d_small, &p->d_aln, &p->d_jobs})
        if (b->p) (void)hipFree(b->p);
    delete p;
}

void ioc_poa_bind(ioc_poa* p, ioc_consensus_ops* ops)
{
    if (!p || !ops) return;
    ops->user = p;
    ops->create = op_create;
    ops->size = op_size;
    ops->add = op_add;
    ops->consensus = op_consensus;
    ops->purge = op_purge;
    ops->spec = getenv("IOC_CONS_SPECULATE") && atoi(getenv("IOC_CONS_SPECULATE")) == 0 ? nullptr : &g_spec_ops;
}

int ioc_poa_graph_export(ioc_poa* p, int side, int idx, int32_t* n_nodes, int32_t* n_edges, char* bases, int32_t* rank,
                         int32_t* edge_from, int32_t* edge_to, int64_t* edge_w)
{
    if (!p || side < 0 || side > 1) return IOC_ERR_ARG;
    {
        int fr = poa_flush(p);
        if (fr != IOC_OK) return fr;
    }
    auto it = p->g[side].find(idx);
    if (it == p->g[side].end()) return IOC_ERR_ARG;
    it->second.ensure();
    const PGraph& G = it->second;
    if (n_nodes) *n_nodes = int32_t(G.n_nodes());
    if (n_edges) *n_edges = int32_t(G.n_edges());
    for (size_t i = 0; i < G.n_nodes(); ++i) {
        if (bases) bases[i] = G.base[i];
        if (rank) rank[i] = G.rank[i];
    }
    for (size_t i = 0; i < G.n_edges(); ++i) {
        if (edge_from) edge_from[i] = G.e_from[i];
        if (edge_to) edge_to[i] = G.e_to[i];
        if (edge_w) edge_w[i] = G.e_w[i];
    }
    return IOC_OK;
}

// ---- persistence (the .cer batch files carry one graph per cluster, src/serialize.h:21,37) ----------------------
// blob: "IOCPOA2\0" (round 4 changed what an edge weighs — w[i - 1] + w[i], spoa's — and the order nodes are created in: a graph of
// the "IOCPOA1" builds would take later reads' edges at twice the weight of its own and give a consensus neither build would; it is
// refused by name) | i32 nseq | i32 n_nodes | i32 n_edges | n_nodes x (u8 base, i32 n_aligned, n_aligned x i32) |
// n_edges x (i32 from, i32 to, i64 weight).  The layout is this build's own (spoa's cereal layout is not in the tree).
int64_t ioc_poa_graph_save(ioc_poa* p, int side, int idx, uint8_t* out, int64_t cap)
{
    if (!p || side < 0 || side > 1) return IOC_ERR_ARG;
    {
        int fr = poa_flush(p);
        if (fr != IOC_OK) return int64_t(fr);
    }
    auto it = p->g[side].find(idx);
    if (it == p->g[side].end()) return IOC_ERR_ARG;
    it->second.ensure();
    const PGraph& G = it->second;
    // (written in place: the command line saves thousands of graphs of thousands of nodes)
    const int64_t need = 8 + 12 + int64_t(G.n_nodes()) * 5 + int64_t(G.al_val.size()) * 4 + int64_t(G.n_edges()) * 16;
    if (!out) return need;
    if (cap < need) return IOC_ERR_CAPACITY;
    uint8_t* w = out;
    auto put = [&](const void* v, size_t n) {
        memcpy(w, v, n);
        w += n;
    };
    auto put32 = [&](int32_t v) { put(&v, 4); };
    put("IOCPOA2", 8);
    put32(G.nseq);
    put32(int32_t(G.n_nodes()));
    put32(int32_t(G.n_edges()));
    for (size_t v = 0; v < G.n_nodes(); ++v) {
        *w++ = uint8_t(G.base[v]);
        uint8_t* cnt = w;
        w += 4;
        int32_t na = 0;
        for (int q = G.al_head[v]; q >= 0; q = G.al_next[size_t(q)], ++na) put32(G.al_val[size_t(q)]);
        memcpy(cnt, &na, 4);
    }
    for (size_t e = 0; e < G.n_edges(); ++e) {
        put32(G.e_from[e]);
        put32(G.e_to[e]);
        put(&G.e_w[e], 8);
    }
    if (w - out != need) return ioc_fail(p->ctx, IOC_ERR_STATE, "POA: graph blob size mismatch");
    return need;
}

// blob -> graph (pure host code, no shared state: safe on worker threads); 0, or 1 foreign blob / 2 corrupt / 3 cyclic
static int parse_graph_blob(const uint8_t* in, int64_t len, PGraph& G)
{
    if (!in || len < 20) return 2;  // (magic + three counts; a negative length would put `e` in front of `q`)
    const uint8_t* q = in;
    const uint8_t* e = in + len;
    bool ok = true;
    auto get = [&](void* v, size_t n) {
        if (size_t(e - q) < n) {
            ok = false;
            memset(v, 0, n);
            return;
        }
        memcpy(v, q, n);
        q += n;
    };
    auto get32 = [&]() {
        int32_t v = 0;
        get(&v, 4);
        return v;
    };
    char magic[8];
    get(magic, 8);
    if (ok && memcmp(magic, "IOCPOA1", 8) == 0) return 4;
    if (!ok || memcmp(magic, "IOCPOA2", 8) != 0) return 1;
    G.nseq = get32();
    const int32_t nn = get32(), ne = get32();
    if (!ok || nn < 0 || ne < 0) return 2;
    if (int64_t(nn) > len || int64_t(ne) > len) return 2;  // (a node takes 5 bytes at least)
    for (int32_t i = 0; i < nn; ++i) G.add_node(0);
    for (int32_t i = 0; ok && i < nn; ++i) {
        get(&G.base[size_t(i)], 1);
        const int32_t na = get32();
        if (na < 0 || na > nn) ok = false;
        for (int32_t x = 0; ok && x < na; ++x) {
            const int32_t a = get32();
            if (a < 0 || a >= nn) {
                ok = false;
                break;
            }
            G.al_push(i, a);
        }
    }
    for (int32_t i = 0; ok && i < ne; ++i) {
        const int32_t from = get32(), to = get32();
        int64_t w = 0;
        get(&w, 8);
        if (!ok || from < 0 || from >= nn || to < 0 || to >= nn) {
            ok = false;
            break;
        }
        G.link_edge(from, to, w);
    }
    if (!ok) return 2;
    G.toposort();
    if (G.rank.size() != G.n_nodes()) return 3;
    return 0;
}
static int graph_blob_error(ioc_poa* p, int code)
{
    return ioc_fail(p->ctx, IOC_ERR_INPUT, code == 1   ? "not a graph written by this build"
                                           : code == 3 ? "graph with a cycle"
                                           : code == 4 ? "consensus graph of an earlier build (IOCPOA1: other edge weights and node order); cluster the batch again with this build"
                                                       : "corrupt graph");
}

int ioc_poa_graph_load(ioc_poa* p, int side, int idx, const uint8_t* in, int64_t len)
{
    if (!p || side < 0 || side > 1 || !in || idx < 0 || len < 0) return IOC_ERR_ARG;
    PGraph G;
    const int code = parse_graph_blob(in, len, G);
    if (code) return graph_blob_error(p, code);
    p->g[side][idx] = std::move(G);
    return IOC_OK;
}

// many graphs at once: the blobs are parsed and ordered on the host's cores (a merge of two 1500-cluster batches loads 3000
// graphs of thousands of nodes: 0.8 s one after the other).  Nothing is loaded when one of them is refused.
int ioc_poa_graph_load_many(ioc_poa* p, int side, int32_t count, const int32_t* idx, const uint8_t* const* in, const int64_t* len)
{
    if (!p || side < 0 || side > 1 || count < 0 || (count > 0 && (!idx || !in || !len))) return IOC_ERR_ARG;
    std::vector<PGraph> gs(static_cast<size_t>(count));
    std::vector<int> code(static_cast<size_t>(count), 0);
    for (int32_t x = 0; x < count; ++x)
        if (!in[x] || idx[x] < 0 || len[x] < 0) return IOC_ERR_ARG;
    ioc_parallel_for(size_t(count), [&](size_t x) { code[x] = parse_graph_blob(in[x], len[x], gs[x]); }, 8);
    for (int32_t x = 0; x < count; ++x)
        if (code[size_t(x)]) return graph_blob_error(p, code[size_t(x)]);
    for (int32_t x = 0; x < count; ++x) p->g[side][idx[x]] = std::move(gs[size_t(x)]);
    return IOC_OK;
}

int ioc_poa_last_alignment(ioc_poa* p, int32_t cap, int32_t* nodes, int32_t* pos, int32_t* score)
{
    if (!p) return IOC_ERR_ARG;
    {
        int fr = poa_flush(p);
        if (fr != IOC_OK) return fr;
    }
    const int n = int(p->last_node.size());
    if (score) *score = p->last_score;
    if (nodes && pos) {
        if (cap < n) return IOC_ERR_CAPACITY;
        for (int i = 0; i < n; ++i) {
            nodes[i] = p->last_node[size_t(i)];
            pos[i] = p->last_pos[size_t(i)];
        }
    }
    return n;
}

}  // extern "C"
