// ioc_resolve.hip — the resolve of the read->cluster assignment path on CDNA4 (gfx950): getBestClusterMapping + getMappedRatio
// (src/cluster.cpp:324-406) as sweeps of a fixed point — k_gap_bounds (upper bound of totalMapped), k_decide_scan, k_eval,
// k_decide_pick — and the owner masks of the sharded merge.  Split out of ioc_kernels.hip in round 4; device helpers in
// ioc_kdev.h.
#include "ioc_kdev.h"

// =====================================================================================================
// Resolve: getBestClusterMapping + getMappedRatio (src/cluster.cpp:324-406) for every query, given
// the current guess of which queries open clusters (valid_in).  Three wide kernels per sweep:
//   k_decide_scan  per query: top = max Size over candidates that ARE clusters; top < MinShared ->
//                  new cluster; cut = int(double(top) * MinFraction); every cluster candidate with
//                  int(Size) >= cut whose totalMapped is not cached yet goes to a global work queue;
//   k_eval         one workgroup per queued (query, candidate): totalMapped (decision-independent,
//                  cached in cand_mapped);
//   k_decide_pick  per query: winner = passing candidate of maximal Size (= the first passing one
//                  in descending-Size order); >= 2 passing at that Size -> order-dependent tie flag.
// =====================================================================================================
// ---- an upper bound of totalMapped that needs no walk over the minimizers --------------------------------------------------
// totalMapped (src/cluster.cpp:324-353) adds the distance of two consecutive hits when fewer than limEx query minimizers lie
// between them without a hit, the position of the first hit when its index is < limEx, and the distance of the last hit from the
// end when fewer than limEx minimizers follow it.  With H hits (= the candidate's Size: one hit per query minimizer whose value
// the target holds) there are H - 1 gaps, and a gap that counts spans at most D(limEx) = max_i pos[i + limEx] - pos[i]:
//     totalMapped <= (H - 1) * D + max_{i < limEx} pos[i] + max_{i >= M - limEx} (hpcLen - pos[i]).
// limEx depends on the query's and the target's error cells only (15 x 15 table), so k_gap_bounds leaves (D, head + tail) per
// (query, strand, target cell) and the sweeps reject a candidate whose bound is below the query's threshold without queueing
// its evaluation (IOC_MAPPED_REJECTED in the cache: "evaluated, fails").  Unrelated reads share ~M^2 / 4*3^(k-1) minimizers by
// chance (90 of 4000 at k = 11): enough to be candidates of every query that opens a cluster, never enough to pass.
#define IOC_MAPPED_REJECTED 0xFFFFFFFEu
// maximum of an unsigned value over the 64 lanes of a wave, uniform result (0 is the identity the DPP moves fold away with)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_or_zero<0x111>(v));
    v = max(v, dpp_or_zero<0x112>(v));
    v = max(v, dpp_or_zero<0x114>(v));
    v = max(v, dpp_or_zero<0x118>(v));
    v = max(v, dpp_or_zero<0x142, 0xa>(v));
    v = max(v, dpp_or_zero<0x143, 0xc>(v));
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_gap_bounds(int n, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev, const uint32_t* __restrict__ pos,
             const uint32_t* __restrict__ hpc_len, const uint8_t* __restrict__ err_cell, const int32_t* __restrict__ glim,
             uint2* __restrict__ out, const uint32_t* __restrict__ min_total, uint32_t keep, uint32_t* __restrict__ keep_q)
{
    // one pass over the positions of a strand serves the 15 target cells at once: per cell three running maxima per thread
    // (widest span of limEx consecutive minimizers, farthest head, longest tail), reduced once at the end
    __shared__ uint32_t red[3][15][IOC_WAVES];
    constexpr uint32_t GB_CHUNK = 4096, GB_HALO = 64;
    __shared__ uint32_t s_pos[GB_CHUNK + GB_HALO];
    __shared__ uint32_t s_nonmono;
    const int j = blockIdx.x;
    if (j >= n) return;
    if (threadIdx.x == 0) s_nonmono = 0u;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const int ecr = int(err_cell[j]) - 1;
    const uint32_t hl = hpc_len[j];
    // smallest Size that passes SOME bound of this query (any strand, any target cell): candidates below it never pass
    // (keep_q: fast mode only — the tie sets of the alignment fallback are made of candidates that fail the mapping)
    const uint32_t need = min_total ? min_total[j] : 0u;
    uint32_t smin = 0xFFFFFFFFu;
    uint32_t lim[15];
#pragma unroll
    for (int e = 0; e < 15; ++e) lim[e] = ecr >= 0 ? uint32_t(glim[e * 15 + ecr] + 1) : 0u;
    for (int s = 0; s < 2; ++s) {
        const int64_t b = s ? off_rev[j] : off_fwd[j];
        const uint32_t M = uint32_t((s ? off_rev[j + 1] : off_fwd[j + 1]) - b);
        const uint32_t* p = pos + b;
        uint32_t d[15], hd[15], tl[15];
#pragma unroll
        for (int e = 0; e < 15; ++e) d[e] = hd[e] = tl[e] = 0;
        if (ecr >= 0 && M > 0) {
            // the widest span of lim[e] consecutive minimizers: the limits of a column of the table ascend with the target's cell,
            // equal neighbours (6 - 13 distinct values of 15) share their maximum
            // (the positions pass through LDS, GB_CHUNK at a time with a halo of GB_HALO behind them: the 6 - 13 reads per entry
            // are latency in global memory; a limit above the halo — none in the reference's table — reads global memory)
            for (uint32_t c0 = 0; c0 < M; c0 += GB_CHUNK) {
                const uint32_t cn = (M - c0 < GB_CHUNK + GB_HALO) ? M - c0 : GB_CHUNK + GB_HALO;  // staged entries
                __syncthreads();
                for (uint32_t x = threadIdx.x; x < cn; x += IOC_BLOCK) {
                    const uint32_t v = p[c0 + x];
                    s_pos[x] = v;
                    if (c0 + x + 1u < M && p[c0 + x + 1u] < v) s_nonmono = 1u;  // (a list that does not ascend: no bound for this query)
                }
                const uint32_t ce = (M - c0 < GB_CHUNK) ? M - c0 : GB_CHUNK;
                const bool in_lds = lim[14] <= GB_HALO;
                // (the list's end inside the staged stretch: the last position repeated behind it, so that "the minimizer lim
                // places on, or the last one" is a plain read)
                if (c0 + cn == M && in_lds)
                    for (uint32_t x = cn + threadIdx.x; x < ce + GB_HALO; x += IOC_BLOCK) s_pos[x] = p[M - 1u];
                __syncthreads();
                for (uint32_t x = threadIdx.x; x < ce; x += IOC_BLOCK) {
                    const uint32_t a0 = s_pos[x];
                    const uint32_t left = M - 1u - (c0 + x);  // minimizers behind this one
#pragma unroll
                    for (int e = 0; e < 15; ++e) {
                        if (e > 0 && lim[e] == lim[e - 1]) continue;  // (uniform)
                        uint32_t a1;
                        if (in_lds) {
                            a1 = s_pos[x + lim[e]];
                        } else {
                            const uint32_t st = lim[e] < left ? lim[e] : left;
                            a1 = p[c0 + x + st];
                        }
                        // (positions ascend; a list that does not is flagged above and gets no bound at all, so a wrapped
                        // difference only ever makes a bound that is ignored)
                        const uint32_t span = a1 - a0;
                        d[e] = span > d[e] ? span : d[e];
                    }
                }
            }
#pragma unroll
            for (int e = 1; e < 15; ++e)
                if (lim[e] == lim[e - 1]) d[e] = d[e - 1];
            // the farthest head (a first hit at index < lim still counts its position) and the longest tail (a last hit with fewer
            // than lim minimizers behind it still counts the rest of the sequence): the first / last lim[14] entries
            const uint32_t lmax = lim[14] < M ? lim[14] : M;
            for (uint32_t i = threadIdx.x; i < lmax; i += IOC_BLOCK) {
                const uint32_t a0 = p[i], a1 = p[M - 1u - i];
                const uint32_t t1 = hl > a1 ? hl - a1 : 0u;
#pragma unroll
                for (int e = 0; e < 15; ++e)
                    if (i < lim[e]) {
                        hd[e] = a0 > hd[e] ? a0 : hd[e];
                        tl[e] = t1 > tl[e] ? t1 : tl[e];
                    }
            }
        }
        // (the wave's maxima by DPP — row_shr 1 2 4 8, row_bcast 15 / 31, the result in lane 63 —: 45 values through six
        // ds_bpermute rounds each were as long as the pass over the positions)
#pragma unroll
        for (int e = 0; e < 15; ++e) {
            const uint32_t x = wave_max_u32(d[e]), y = wave_max_u32(hd[e]), z = wave_max_u32(tl[e]);
            if (lane == 0) {
                red[0][e][wave] = x;
                red[1][e][wave] = y;
                red[2][e][wave] = z;
            }
        }
        __syncthreads();
        if (threadIdx.x < 15) {
            const int e = threadIdx.x;
            uint32_t D = 0, HD = 0, TL = 0;
            for (int w = 0; w < IOC_WAVES; ++w) {
                D = red[0][e][w] > D ? red[0][e][w] : D;
                HD = red[1][e][w] > HD ? red[1][e][w] : HD;
                TL = red[2][e][w] > TL ? red[2][e][w] : TL;
            }
            uint2 r = make_uint2(0u, 0u);
            uint32_t thr = 0xFFFFFFFFu;
            if (ecr >= 0 && M > 0) {
                r = make_uint2(D, HD + TL);
                // (Size - 1) * D + HT >= need  <=>  Size >= ceil((need - HT) / D) + 1
                const uint32_t ht = HD + TL;
                thr = need <= ht ? 0u : (D ? (need - ht + D - 1u) / D + 1u : 0xFFFFFFFFu);
                if (s_nonmono) {  // the spans above assume ascending positions (the extractor's lists do): a bound that rejects nothing
                    r = make_uint2(0u, 0xFFFFFFFFu);
                    thr = 0u;
                }
            }
            out[(size_t(j) * 2 + size_t(s)) * 15 + size_t(e)] = r;
            // minimum over the 15 cells (lanes 0..14 of wave 0)
            for (int o = 8; o > 0; o >>= 1) {
                const uint32_t t = __shfl_down(thr, o);
                if (lane + o < 15) thr = t < thr ? t : thr;
            }
            if (threadIdx.x == 0) smin = thr < smin ? thr : smin;
        }
        __syncthreads();
    }
    if (keep_q && threadIdx.x == 0) keep_q[j] = (ecr >= 0 && smin != 0xFFFFFFFFu && smin > keep) ? smin : keep;
}

// true: the candidate (key, Size sz) of query j cannot reach `need` (see k_gap_bounds)
__device__ __forceinline__ bool bound_rejects(const DecideArgs& a, int j, uint32_t key, uint32_t sz, uint32_t need)
{
    const uint32_t tg = key >> 1;
    const int ecl = (tg < a.L ? int(a.left_err[tg]) : int(a.err_cell[tg - a.L])) - 1;
    if (ecl < 0 || sz == 0) return false;
    const uint2 b = a.gap_bound[(size_t(j) * 2 + size_t(key & 1u)) * 15 + size_t(ecl)];
    const unsigned long long B = (unsigned long long)(sz - 1u) * b.x + b.y;
    return B < (unsigned long long)need;
}

#define IOC_CUT_NEG INT32_MAX
#define IOC_BITWORDS 256    // 16384 minimizers per strand per pass (slow path)
#define IOC_EVAL_ILP 8

// (round 4) ONE pass over a query's candidate list per phase: the candidates of the first 2048 entries stay in registers between
// the maximum and the selection, the selected ones — a handful: the list's median length is 12, its mean 900, and what passes
// the Size rule is 1 - 3 — are staged in LDS, queued with one atomic, and left as the query's WALK for k_decide_pick, which then
// reads those instead of the list.  Before: three passes here and one in k_decide_pick over (key, Size, cached totalMapped) of
// 2.7 M candidates per sweep, 0.33 of the fast step's 0.77 ms of resolve.
#define IOC_SCAN_CACHE 8      // candidates per thread kept in registers (phase 1)
#define IOC_SCAN_ITEMS 256    // items staged per query before they go to the queue one by one
__global__ void __launch_bounds__(IOC_BLOCK)
k_decide_scan(DecideArgs a)
{
    __shared__ uint32_t red[IOC_WAVES];
    __shared__ uint32_t s_top, s_base, s_nw, s_ni;
    __shared__ uint32_t s_walk[IOC_WALK_SLOTS], s_item[IOC_SCAN_ITEMS];
    const int j = owned_from(a.first, int(blockIdx.x), a.own_stride, a.own_offset);  // (sharded merge: this rank's queries)
    if (j >= a.n) return;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t L = a.L;
    const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
    const uint32_t C = a.cand_count[j];
    int cut;
    uint32_t top;
    uint32_t csz[IOC_SCAN_CACHE];  // phase 1: Size of candidate k * IOC_BLOCK + threadIdx.x if it is a cluster, else 0
    if (threadIdx.x == 0) {
        s_nw = 0;
        s_ni = 0;
    }
    if (a.phase == 1) {
        if (a.forced_t[j] != INT32_MIN) {
            if (threadIdx.x == 0) {
                a.cut[j] = IOC_CUT_NEG;
                a.walk_n[j] = 0;
            }
            return;
        }
        top = 0;
#pragma unroll
        for (int k = 0; k < IOC_SCAN_CACHE; ++k) {
            const uint32_t c = uint32_t(k) * IOC_BLOCK + threadIdx.x;
            uint32_t v = 0;
            if (c < C) {
                const uint32_t tg = a.cand_key[cbase + c] >> 1;
                const bool ok = (tg < L) || a.valid_in[tg - L];
                v = ok ? a.cand_size[cbase + c] : 0u;
            }
            csz[k] = v;
            top = v > top ? v : top;
        }
        for (uint32_t c = IOC_SCAN_CACHE * IOC_BLOCK + threadIdx.x; c < C; c += IOC_BLOCK) {
            const uint32_t tg = a.cand_key[cbase + c] >> 1;
            const bool ok = (tg < L) || a.valid_in[tg - L];
            const uint32_t sz = a.cand_size[cbase + c];
            if (ok && sz > top) top = sz;
        }
        top = wave_max_u32(top);
        if (lane == 0) red[wave] = top;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < IOC_WAVES; ++w) t = red[w] > t ? red[w] : t;
            s_top = t;
        }
        __syncthreads();
        top = s_top;
        if (top < uint32_t(a.min_shared)) {
            if (threadIdx.x == 0) {
                a.cut[j] = IOC_CUT_NEG;
                a.walk_n[j] = 0;
            }
            return;
        }
        cut = int(double(top) * a.min_fraction);
        if (threadIdx.x == 0) {
            a.cut[j] = cut;
            a.top[j] = top;
        }
    } else {
        if (a.done[j]) return;
        cut = a.cut[j];
        top = a.top[j];
        if (cut == IOC_CUT_NEG) return;
        __syncthreads();  // (s_nw / s_ni are zero)
    }
    // phase 1 walks only the candidates of maximal Size (the first ones the reference walks): most queries are decided by them;
    // phase 2 the rest of the walk for the undecided queries.  A walk candidate whose totalMapped is not cached yet — and that its
    // upper bound does not reject — is an item for k_eval.
    const uint32_t need = a.min_total[j];
    auto take = [&](uint32_t c, uint32_t sz) {  // candidate c is a cluster and passes the Size rule of the phase
        const uint32_t pw = atomicAdd(&s_nw, 1u);
        if (pw < IOC_WALK_SLOTS) s_walk[pw] = c;
        if (a.cand_mapped[cbase + c] != 0xFFFFFFFFu) return;
        if (a.gap_bound && bound_rejects(a, j, a.cand_key[cbase + c], sz, need)) {
            a.cand_mapped[cbase + c] = IOC_MAPPED_REJECTED;
            return;
        }
        const uint32_t pi = atomicAdd(&s_ni, 1u);
        if (pi < IOC_SCAN_ITEMS) {
            s_item[pi] = c;
        } else {  // (a query with hundreds of unevaluated candidates: its further items go to the queue one by one)
            const uint32_t slot = atomicAdd(a.q_count, 1u);
            if (slot < a.q_cap) {
                a.q_items[2 * size_t(slot)] = uint32_t(j);
                a.q_items[2 * size_t(slot) + 1] = c;
            }
        }
    };
    if (a.phase == 1) {
#pragma unroll
        for (int k = 0; k < IOC_SCAN_CACHE; ++k)
            if (csz[k] == top) take(uint32_t(k) * IOC_BLOCK + threadIdx.x, top);  // (top >= MinShared > 0: never an empty slot)
        for (uint32_t c = IOC_SCAN_CACHE * IOC_BLOCK + threadIdx.x; c < C; c += IOC_BLOCK) {
            const uint32_t sz = a.cand_size[cbase + c];
            if (sz != top) continue;
            const uint32_t tg = a.cand_key[cbase + c] >> 1;
            if ((tg < L) || a.valid_in[tg - L]) take(c, sz);
        }
    } else {
        for (uint32_t c = threadIdx.x; c < C; c += IOC_BLOCK) {
            const uint32_t sz = a.cand_size[cbase + c];
            if (int(sz) < cut) continue;
            const uint32_t tg = a.cand_key[cbase + c] >> 1;
            if ((tg < L) || a.valid_in[tg - L]) take(c, sz);
        }
    }
    __syncthreads();
    const uint32_t nw = s_nw, ni = s_ni < IOC_SCAN_ITEMS ? s_ni : IOC_SCAN_ITEMS;
    if (threadIdx.x == 0) {
        a.walk_n[j] = nw <= IOC_WALK_SLOTS ? nw : IOC_WALK_OVERFLOW;
        s_base = ni ? atomicAdd(a.q_count, ni) : 0u;
    }
    if (threadIdx.x < nw && threadIdx.x < IOC_WALK_SLOTS) a.walk_c[size_t(j) * IOC_WALK_SLOTS + threadIdx.x] = s_walk[threadIdx.x];
    __syncthreads();
    // a query's staged items occupy one contiguous range of the queue (k_eval reuses the query's minimizers across consecutive
    // items)
    for (uint32_t x = threadIdx.x; x < ni; x += IOC_BLOCK) {
        const uint32_t slot = s_base + x;
        if (slot < a.q_cap) {
            a.q_items[2 * size_t(slot)] = uint32_t(j);
            a.q_items[2 * size_t(slot) + 1] = s_item[x];
        }
    }
}

// totalMapped of one (query, target, strand): src/cluster.cpp:324-353 with the pow() predicate
// replaced by the integer gap limit (a gap of n missing minimizers passes iff n < limEx).
// Slow path of one evaluation (target sets above 4096 values, i.e. reads beyond ~13 kb HPC as
// representatives): membership by a branchless binary search in the sorted set in global memory.
__device__ __forceinline__ bool set_contains_global(const uint32_t* __restrict__ set, uint32_t setN, uint32_t hp2,
                                                    uint32_t v)
{
    uint32_t pos = 0;
    for (uint32_t h = hp2; h > 0; h >>= 1) {
        const uint32_t q = pos + h;
        if (q <= setN && set[q - 1] < v) pos = q;
    }
    return pos < setN && set[pos] == v;
}

__device__ __forceinline__ uint32_t eval_total_mapped(const uint32_t* __restrict__ qmin,
                                                      const uint32_t* __restrict__ qpos, uint32_t M,
                                                      const uint32_t* set, uint32_t setN, uint32_t limEx, uint32_t hpcLen, unsigned long long* bits,
                                                      uint32_t* red, uint32_t* carry, unsigned long long* diag)
{
    const int lane = lane_id(), wave = wave_id();
    uint32_t total = 0;
    long long ta = 0, tb = 0, tc = 0;
    if (threadIdx.x == 0) {
        carry[0] = 0;  // any hit so far
        carry[1] = 0;  // index of the last hit so far
    }
    uint32_t hp2 = 1;
    while ((hp2 << 1) <= setN) hp2 <<= 1;
    if (setN == 0) hp2 = 0;
    __syncthreads();
    for (uint32_t pbase = 0; pbase < M; pbase += IOC_BITWORDS * 64) {
        const uint32_t Mp = (M - pbase < IOC_BITWORDS * 64) ? (M - pbase) : IOC_BITWORDS * 64;
        const uint32_t nwords = (Mp + 63) >> 6;
        if (diag) ta = clock64();
        // phase A: hit bitmap, one 64-bit word per wave step; the loads of IOC_EVAL_ILP words are
        // issued together (coalesced reads of qmin)
        for (uint32_t wd0 = wave * IOC_EVAL_ILP; wd0 < nwords; wd0 += IOC_WAVES * IOC_EVAL_ILP) {
            uint32_t v[IOC_EVAL_ILP];
            bool in[IOC_EVAL_ILP];
#pragma unroll
            for (int u = 0; u < IOC_EVAL_ILP; ++u) {
                const uint32_t i = pbase + (wd0 + u) * 64 + lane;
                in[u] = (wd0 + u < nwords) && (i < M);
                v[u] = in[u] ? qmin[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < IOC_EVAL_ILP; ++u) {
                const bool hit = in[u] && set_contains_global(set, setN, hp2, v[u]);
                const unsigned long long m = __ballot(hit);
                if (lane == 0 && wd0 + u < nwords) bits[wd0 + u] = m;
            }
        }
        __syncthreads();
        if (diag) tb = clock64();
        // phase B: one thread per minimizer index (coalesced reads of qpos); the previous hit is
        // the highest set bit below i: same word, else an earlier word, else the carry of the
        // previous pass.
        uint32_t local = 0;
        const uint32_t had_any = carry[0], had_last = carry[1];
        for (uint32_t ii = threadIdx.x; ii < Mp; ii += IOC_BLOCK) {
            const uint32_t wd = ii >> 6, bit = ii & 63u;
            const unsigned long long m = bits[wd];
            if (!((m >> bit) & 1ull)) continue;
            const uint32_t i = pbase + ii;
            bool pany = false;
            uint32_t pidx = 0;
            const unsigned long long below = m & ((1ull << bit) - 1ull);
            if (below) {
                pany = true;
                pidx = pbase + wd * 64 + uint32_t(63 - __builtin_clzll(below));
            } else {
                for (int x = int(wd) - 1; x >= 0; --x) {
                    const unsigned long long pm = bits[x];
                    if (pm) {
                        pany = true;
                        pidx = pbase + uint32_t(x) * 64 + uint32_t(63 - __builtin_clzll(pm));
                        break;
                    }
                }
                if (!pany && had_any) {
                    pany = true;
                    pidx = had_last;
                }
            }
            if (!pany) {
                if (i < limEx) local += qpos[i];  // pow(pError, hits[0].Index) >= p0
            } else if (i - pidx - 1 < limEx) {
                local += qpos[i] - qpos[pidx];
            }
        }
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
        if (lane == 0) red[wave] = local;
        __syncthreads();
        if (diag && threadIdx.x == 0) {
            tc = clock64();
            atomicAdd(&diag[5], (unsigned long long)(tb - ta));
            atomicAdd(&diag[6], (unsigned long long)(tc - tb));
        }
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < IOC_WAVES; ++w) sum += red[w];
            red[IOC_WAVES] = sum;
            for (int x = int(nwords) - 1; x >= 0; --x) {
                const unsigned long long pm = bits[x];
                if (pm) {
                    carry[0] = 1;
                    carry[1] = pbase + uint32_t(x) * 64 + uint32_t(63 - __builtin_clzll(pm));
                    break;
                }
            }
        }
        __syncthreads();
        total += red[IOC_WAVES];
        __syncthreads();
    }
    // tail: pow(pError, nMins - (lastIdx + 1)) >= p0
    const uint32_t any = carry[0], last = carry[1];
    if (any && (M - last - 1 < limEx)) total += hpcLen - qpos[last];
    __syncthreads();
    return total;
}

// Fast path of one evaluation (target set <= 4096 values, the common case).
//   * every global load is issued up front and coalesced (16 set values per thread; 16 query values per
//     thread and pass, kept in registers across consecutive items of the same (query, strand));
//   * membership = an open-addressed hash table of the target's set in LDS (8192 slots for <= 4096 values: one
//     ds_cmpst per set value, 1.3 ds_read per query value on average, all batched).  Round 3 had a 64 Kbit filter in front
//     of a binary search in an LDS copy of the sorted set: the candidates that get evaluated are the RELATED ones — three
//     query values in four are members —, so the filter filtered little and the 12 dependent reads of the search per
//     positive were 21 of an evaluation's 47 us (IOC_EVAL_DIAG);
//   * hit bitmap -> previous-hit table -> gap scan, a thread per 16 indices: the positions of its hits are requested
//     together (one latency instead of one per hit: 16 of the 47 us).
#define IOC_EV_PER 16                              // indices per thread per pass
#define IOC_EV_PASS (IOC_EV_PER * IOC_BLOCK)       // 4096 query minimizers per pass
#define IOC_EV_HBITS 13
#define IOC_EV_HSLOTS (1u << IOC_EV_HBITS)         // 8192 slots: load <= 0.5
struct EvQuery {
    uint32_t qv[IOC_EV_PER];
    uint32_t pend0;  // valid-index mask of the cached pass
};
struct EvLds {
    __attribute__((aligned(16))) uint32_t htab[IOC_EV_HSLOTS];  // the target's set (IOC_EMPTY: free slot)
    unsigned long long bits[64];            // hit bitmap of the pass
    uint32_t prevlast[65];
    uint32_t red[IOC_WAVES + 1];
    uint32_t carry[4];
    uint32_t nhits;
    uint32_t has_empty;                     // the set holds the value IOC_EMPTY itself (k = 16: sixteen T)
};

__device__ __forceinline__ uint32_t ev_hash(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - IOC_EV_HBITS); }

__device__ __forceinline__ uint32_t eval_fast(const uint32_t* __restrict__ qmin, const uint32_t* __restrict__ qpos,
                                              uint32_t M, const uint32_t* __restrict__ set, uint32_t setN,
                                              uint32_t limEx, uint32_t hpcLen, EvLds& S, EvQuery& Q, bool reuse,
                                              unsigned long long* diag)
{
    long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    if (diag) s0 = clock64();
    const int lane = lane_id(), wave = wave_id();
    uint32_t sv[IOC_EV_PER];
#pragma unroll
    for (int u = 0; u < IOC_EV_PER; ++u) {
        const uint32_t i = uint32_t(u) * IOC_BLOCK + threadIdx.x;
        sv[u] = i < setN ? set[i] : IOC_EMPTY;
    }
    const bool single = M <= IOC_EV_PASS;
    if (!(reuse && single)) {
        Q.pend0 = 0;
#pragma unroll
        for (int u = 0; u < IOC_EV_PER; ++u) {
            const uint32_t li = (uint32_t(u) * IOC_WAVES + wave) * 64 + lane;
            const bool in = li < M;
            Q.qv[u] = in ? qmin[li] : 0u;
            if (in) Q.pend0 |= 1u << u;
        }
    }
    {
        uint4* h4 = reinterpret_cast<uint4*>(S.htab);
        const uint4 e4 = make_uint4(IOC_EMPTY, IOC_EMPTY, IOC_EMPTY, IOC_EMPTY);
#pragma unroll
        for (uint32_t i = 0; i < IOC_EV_HSLOTS / 4 / IOC_BLOCK; ++i) h4[i * IOC_BLOCK + threadIdx.x] = e4;
    }
    if (threadIdx.x == 0) {
        S.carry[0] = 0;  // any hit so far
        S.carry[1] = 0;  // index of the last hit so far
        S.has_empty = 0;
    }
    __syncthreads();
    if (diag) s1 = clock64();
    {   // insert: the first slot of all 16 values at once, then the (few) values whose slot was taken walk on
        uint32_t hs[IOC_EV_PER], old[IOC_EV_PER];
#pragma unroll
        for (int u = 0; u < IOC_EV_PER; ++u) {
            hs[u] = ev_hash(sv[u]);
            old[u] = sv[u] != IOC_EMPTY ? atomicCAS(&S.htab[hs[u]], IOC_EMPTY, sv[u]) : IOC_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < IOC_EV_PER; ++u) {
            uint32_t h = hs[u], o = old[u];
            while (o != IOC_EMPTY) {  // (set values are distinct: a taken slot holds another value)
                h = (h + 1u) & (IOC_EV_HSLOTS - 1u);
                o = atomicCAS(&S.htab[h], IOC_EMPTY, sv[u]);
            }
        }
        // (IOC_EMPTY as a VALUE of the set — the sorted set's last entry — cannot live in the table)
        if (setN && threadIdx.x == ((setN - 1u) & (IOC_BLOCK - 1u)) && sv[(setN - 1u) / IOC_BLOCK] == IOC_EMPTY) S.has_empty = 1u;
    }
    __syncthreads();
    if (diag) s2 = clock64();
    const uint32_t has_empty = S.has_empty;
    uint32_t total = 0;
    for (uint32_t pbase = 0; pbase < M; pbase += IOC_EV_PASS) {
        const uint32_t Mp = (M - pbase < IOC_EV_PASS) ? (M - pbase) : IOC_EV_PASS;
        // thread owns local indices (u * IOC_WAVES + wave) * 64 + lane: bit `lane` of word u * IOC_WAVES + wave of the pass
        uint32_t qv[IOC_EV_PER];
        uint32_t pend = 0;
        if (pbase == 0) {
            pend = Q.pend0;
#pragma unroll
            for (int u = 0; u < IOC_EV_PER; ++u) qv[u] = Q.qv[u];
        } else {
#pragma unroll
            for (int u = 0; u < IOC_EV_PER; ++u) {
                const uint32_t li = (uint32_t(u) * IOC_WAVES + wave) * 64 + lane;
                const bool in = li < Mp;
                qv[u] = in ? qmin[pbase + li] : 0u;
                if (in) pend |= 1u << u;
            }
        }
        // membership: the first slot of all 16 values at once; a value is decided by its own slot's content unless another
        // value sits there
        uint32_t hs[IOC_EV_PER], x[IOC_EV_PER];
#pragma unroll
        for (int u = 0; u < IOC_EV_PER; ++u) {
            hs[u] = ev_hash(qv[u]);
            x[u] = S.htab[hs[u]];
        }
#pragma unroll
        for (int u = 0; u < IOC_EV_PER; ++u) {
            uint32_t h = hs[u], y = x[u];
            const uint32_t v = qv[u];
            bool member = false;
            if ((pend >> u) & 1u) {
                if (v == IOC_EMPTY) {
                    member = has_empty != 0u;
                } else {
                    while (y != v && y != IOC_EMPTY) {
                        h = (h + 1u) & (IOC_EV_HSLOTS - 1u);
                        y = S.htab[h];
                    }
                    member = y == v;
                }
            }
            const unsigned long long m = __ballot(member);
            if (lane == 0) S.bits[uint32_t(u) * IOC_WAVES + uint32_t(wave)] = m;
        }
        __syncthreads();
        if (diag) s3 = clock64();
        // prevlast[w] = 1 + local index of the last hit in words < w (0 = none); hit count
        if (wave == 0) {
            const unsigned long long m = S.bits[lane];
            uint32_t v = m ? uint32_t(lane) * 64 + uint32_t(63 - __builtin_clzll(m)) + 1u : 0u;
            uint32_t cnt = uint32_t(__popcll(m));
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(v, o);
                const uint32_t c2 = __shfl_up(cnt, o);
                if (lane >= o) {
                    v = t > v ? t : v;
                    cnt += c2;
                }
            }
            S.prevlast[lane + 1] = v;
            if (lane == 0) S.prevlast[0] = 0;
            if (lane == 63) S.nhits = cnt;
        }
        __syncthreads();
        // ---- gap scan over the hit bitmap: one thread per 16-bit quarter word ----
        uint32_t local = 0;
        const uint32_t had_any = S.carry[0], had_last = S.carry[1];
        {
            const uint32_t wd = threadIdx.x >> 2;        // 64 words
            const uint32_t qtr = threadIdx.x & 3u;       // 16 bits each
            const unsigned long long m = S.bits[wd];
            const uint32_t part = uint32_t(m >> (16 * qtr)) & 0xFFFFu;
            // previous hit before this quarter
            bool pany = false;
            uint32_t pidx = 0;
            const unsigned long long below = qtr ? (m & ((1ull << (16 * qtr)) - 1ull)) : 0ull;
            if (below) {
                pany = true;
                pidx = pbase + wd * 64 + uint32_t(63 - __builtin_clzll(below));
            } else {
                const uint32_t pl = S.prevlast[wd];
                if (pl) {
                    pany = true;
                    pidx = pbase + pl - 1u;
                } else if (had_any) {
                    pany = true;
                    pidx = had_last;
                }
            }
            if (part) {
                // the positions of this quarter's hits and of the hit before it: requested together
                const uint32_t i0 = pbase + wd * 64 + 16 * qtr;
                uint32_t pp[16];
#pragma unroll
                for (int bq = 0; bq < 16; ++bq) pp[bq] = ((part >> bq) & 1u) ? qpos[i0 + uint32_t(bq)] : 0u;
                uint32_t ppos = pany ? qpos[pidx] : 0u;
#pragma unroll
                for (int bq = 0; bq < 16; ++bq) {
                    if (!((part >> bq) & 1u)) continue;
                    const uint32_t i = i0 + uint32_t(bq);
                    if (!pany) {
                        if (i < limEx) local += pp[bq];  // pow(pError, hits[0].Index) >= p0
                    } else if (i - pidx - 1 < limEx) {
                        local += pp[bq] - ppos;
                    }
                    pany = true;
                    pidx = i;
                    ppos = pp[bq];
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
        if (lane == 0) S.red[wave] = local;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < IOC_WAVES; ++w) sum += S.red[w];
            S.red[IOC_WAVES] = sum;
            const uint32_t pl = S.prevlast[64];
            if (pl) {
                S.carry[0] = 1;
                S.carry[1] = pbase + pl - 1u;
            }
        }
        __syncthreads();
        total += S.red[IOC_WAVES];
        __syncthreads();
    }
    // tail: pow(pError, nMins - (lastIdx + 1)) >= p0
    if (S.carry[0] && (M - S.carry[1] - 1 < limEx)) total += hpcLen - qpos[S.carry[1]];
    __syncthreads();
    if (diag && threadIdx.x == 0) {
        s4 = clock64();
        atomicAdd(&diag[5], (unsigned long long)(s1 - s0));  // issue loads + clear the table
        atomicAdd(&diag[6], (unsigned long long)(s2 - s1));  // table build (waits for the loads)
        atomicAdd(&diag[7], (unsigned long long)(s3 - s2));  // membership
        atomicAdd(&diag[0], (unsigned long long)(s4 - s3));  // gap scan + reduce
    }
    return total;
}

// (Round 4 tried the opposite layout — every entry's values hashed ONCE per index build into a table in global memory, one WAVE
// per evaluation probing it, no LDS, 64 registers, 8192 evaluations in flight — and measured it at 691 us of k_eval per fast
// step against 366: 3000 tables of 32 KB are 98 MB, probed 4 bytes at a time at random, each by one or two evaluations only;
// what the workgroup version reads once and coalesced (48 KB per evaluation) became 6000 cache-line requests.  Taken out.)
#ifndef IOC_EVAL_MINWAVES
#define IOC_EVAL_MINWAVES 4  // 128 registers (5 spilled): four workgroups per CU instead of three
#endif
__global__ void __launch_bounds__(IOC_BLOCK, IOC_EVAL_MINWAVES)
k_eval(DecideArgs a)
{
    __shared__ unsigned long long bits[IOC_BITWORDS];  // slow path only
    __shared__ uint32_t red[IOC_WAVES + 1];
    __shared__ uint32_t carry[4];
    __shared__ EvLds S;
    uint32_t count = *a.q_count;
    if (count > a.q_cap) count = a.q_cap;
    const uint32_t L = a.L;
    // each workgroup takes one contiguous chunk of the queue: consecutive items share the query
    const uint32_t per = (count + gridDim.x - 1) / gridDim.x;
    const uint32_t w_begin = blockIdx.x * per;
    const uint32_t w_end = (w_begin + per < count) ? (w_begin + per) : count;
    EvQuery Q;
    Q.pend0 = 0;
    uint32_t prev_j = 0xFFFFFFFFu;
    int prev_strand = -1;
    for (uint32_t w = w_begin; w < w_end; ++w) {
        const uint32_t j = a.q_items[2 * size_t(w)];
        const uint32_t c = a.q_items[2 * size_t(w) + 1];
        const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
        const uint32_t key = a.cand_key[cbase + c];
        const uint32_t tg = key >> 1;
        const int strandbit = int(key & 1u);
        const uint32_t* set;
        uint32_t setN;
        int ecl;
        if (tg < L) {
            set = a.lset_val + a.lset_off[tg];
            setN = uint32_t(a.lset_off[tg + 1] - a.lset_off[tg]);
            ecl = int(a.left_err[tg]) - 1;
        } else {
            set = a.dvals + a.doff[tg - L];
            setN = a.dcount[tg - L];
            ecl = int(a.err_cell[tg - L]) - 1;
        }
        const int ecr = int(a.err_cell[j]) - 1;
        const uint32_t limEx = uint32_t(a.glim[ecl * 15 + ecr] + 1);  // gap n passes iff n < limEx
        const int64_t qb = strandbit ? a.off_rev[j] : a.off_fwd[j];
        const uint32_t M = uint32_t((strandbit ? a.off_rev[j + 1] : a.off_fwd[j + 1]) - qb);
        uint32_t tm;
        long long t1 = 0, t2 = 0;
        if (setN <= IOC_EV_PASS) {
            const bool reuse = (j == prev_j) && (strandbit == prev_strand);
            if (a.diag) t1 = clock64();
            tm = eval_fast(a.mins + qb, a.pos + qb, M, set, setN, limEx, a.hpc_len[j], S, Q, reuse, a.diag);
            prev_j = j;
            prev_strand = strandbit;
        } else {
            tm = eval_total_mapped(a.mins + qb, a.pos + qb, M, set, setN, limEx, a.hpc_len[j], bits, red, carry, a.diag);
            prev_j = 0xFFFFFFFFu;
        }
        if (a.diag && threadIdx.x == 0) {
            t2 = clock64();
            atomicAdd(&a.diag[1], (unsigned long long)(t2 - t1));  // phases A + B
            atomicAdd(&a.diag[2], 1ull);
            atomicAdd(&a.diag[3], (unsigned long long)M);
            atomicAdd(&a.diag[4], (unsigned long long)setN);
        }
        if (threadIdx.x == 0) {
            a.cand_mapped[cbase + c] = tm;
            if (a.n_evals) atomicAdd(a.n_evals, 1ull);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_decide_pick(DecideArgs a)
{
    __shared__ uint32_t rs[IOC_WAVES], re[IOC_WAVES], rc[IOC_WAVES], rm[IOC_WAVES];
    const int j = owned_from(a.first, int(blockIdx.x), a.own_stride, a.own_offset);
    if (j >= a.n) return;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t L = a.L;
    if (a.phase == 2 && a.done[j]) return;
    const int32_t ft = a.forced_t[j];
    if (ft != INT32_MIN) {
        if (threadIdx.x == 0) {
            uint8_t nv = (ft == -1) ? 1 : 0;  // -1 opens a cluster; -2 = excluded entry (gated)
            a.dec_target[j] = ft;
            a.dec_strand[j] = (ft < 0) ? 0 : a.forced_s[j];
            a.flags[j] = 0;
            a.valid_out[j] = nv;
            a.done[j] = 1;
            if (nv != a.valid_in[j]) atomicMin(a.first_changed, uint32_t(j));
        }
        return;
    }
    const int cut = a.cut[j];
    int32_t out_t = -1;
    int8_t out_s = 0;
    uint8_t out_f = 0;
    bool decided = true;
    bool provisional = false;  // a lazy sweep's cluster opener by default: decision written, walk not finished (done stays 0)
    __shared__ uint32_t s_tn, s_tk[IOC_TIE_SLOTS];
    if (a.tie_count) {
        if (threadIdx.x == 0) s_tn = 0;
        __syncthreads();
    }
    if (cut != IOC_CUT_NEG) {
        const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
        const uint32_t C = a.cand_count[j];
        const uint32_t need = a.min_total[j];
        const uint32_t top = a.top[j];
        uint32_t bs = 0, be = 0xFFFFFFFFu, bc = 0, miss = 0;
        // the walk as k_decide_scan of this phase left it (a handful of candidates), or — more than IOC_WALK_SLOTS of them — the list
        const uint32_t wn = a.walk_n[j];
        const bool listed = wn != IOC_WALK_OVERFLOW;
        const uint32_t* wl = a.walk_c + size_t(j) * IOC_WALK_SLOTS;
        const uint32_t n_iter = listed ? wn : C;
        for (uint32_t x = threadIdx.x; x < n_iter; x += IOC_BLOCK) {
            const uint32_t c = listed ? wl[x] : x;
            const uint32_t key = a.cand_key[cbase + c];
            const uint32_t tg = key >> 1;
            const bool ok = (tg < L) || a.valid_in[tg - L];
            const uint32_t sz = a.cand_size[cbase + c];
            if (a.tie_count && ok && sz == top) {  // what getBestClusterAln would try (cluster.cpp:481-489)
                const uint32_t pos = atomicAdd(&s_tn, 1u);
                if (pos < IOC_TIE_SLOTS) s_tk[pos] = key;
            }
            if (!ok || (a.phase == 1 ? sz != top : int(sz) < cut)) continue;
            const uint32_t tm = a.cand_mapped[cbase + c];
            if (tm == IOC_MAPPED_REJECTED) continue;  // fails by its upper bound
            if (tm == 0xFFFFFFFFu) {
                miss = 1;
                continue;
            }
            if (tm >= need) {
                if (sz > bs) {
                    bs = sz;
                    be = c;
                    bc = 1;
                } else if (sz == bs) {
                    bc++;
                    be = c < be ? c : be;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t os = __shfl_down(bs, o), oe = __shfl_down(be, o), oc = __shfl_down(bc, o);
            miss |= __shfl_down(miss, o);
            if (os > bs) {
                bs = os;
                be = oe;
                bc = oc;
            } else if (os == bs) {
                bc += oc;
                be = oe < be ? oe : be;
            }
        }
        if (lane == 0) {
            rs[wave] = bs;
            re[wave] = be;
            rc[wave] = bc;
            rm[wave] = miss;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bs = 0;
            be = 0xFFFFFFFFu;
            bc = 0;
            miss = 0;
            for (int w = 0; w < IOC_WAVES; ++w) {
                miss |= rm[w];
                if (rs[w] > bs) {
                    bs = rs[w];
                    be = re[w];
                    bc = rc[w];
                } else if (rs[w] == bs) {
                    bc += rc[w];
                    be = re[w] < be ? re[w] : be;
                }
            }
            if (miss) atomicAdd(a.incomplete, 1u);
            if (bs > 0 && be != 0xFFFFFFFFu) {
                const uint32_t key = a.cand_key[cbase + be];
                out_t = int32_t(key >> 1);
                out_s = (key & 1u) ? -1 : 1;
                if (bc > 1) out_f |= 1;
            } else {
                out_f |= 2;  // no mapping hit although top >= MinShared (cluster.cpp:553-566)
                if (a.aln_t && a.aln_t[j] != INT32_MIN) {  // the alignment fallback's verdict for this query
                    out_t = a.aln_t[j];
                    out_s = out_t < 0 ? int8_t(0) : a.aln_s[j];
                    if (out_t < 0) out_t = -1;
                }
                // phase 1 only looked at the maximal-Size candidates: the walk goes on in phase 2 —
                // unless this is a lazy sweep, which provisionally lets the query open a cluster (what
                // almost always happens) and leaves the rest of the walk to the final exact sweeps
                if (a.phase == 1 && !a.lazy) decided = false;
                if (a.phase == 1 && a.lazy) provisional = true;
            }
            if (miss) decided = false;
        }
    }
    if (threadIdx.x == 0) {
        if (a.tie_count) {
            a.tie_count[j] = (cut != IOC_CUT_NEG) ? s_tn : 0u;
            for (int t = 0; t < IOC_TIE_SLOTS; ++t) a.tie_keys[size_t(j) * IOC_TIE_SLOTS + t] = s_tk[t];
        }
        a.done[j] = (decided && !provisional) ? 1 : 0;
        if (decided) {
            const uint8_t nv = (out_t < 0) ? 1 : 0;
            a.dec_target[j] = out_t;
            a.dec_strand[j] = out_s;
            a.flags[j] = out_f;
            a.valid_out[j] = nv;
            if (nv != a.valid_in[j]) atomicMin(a.first_changed, uint32_t(j));
        }
    }
}

// First guess of "entry j opens a cluster" for the fixed-point resolve: j probably joins an earlier
// cluster when some earlier entry shares more than ~5 % of its minimizers (background between unrelated
// reads is ~1.6 % at k = 11).  Any guess converges to the same result; a good one saves sweeps and
// evaluations.
__global__ void __launch_bounds__(IOC_BLOCK)
k_guess_valid(int n, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
              const uint32_t* __restrict__ top_all, uint8_t* __restrict__ valid)
{
    const int j = blockIdx.x * IOC_BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint32_t nf = uint32_t(off_fwd[j + 1] - off_fwd[j]), nr = uint32_t(off_rev[j + 1] - off_rev[j]);
    const uint32_t m = nf < nr ? nf : nr;
    valid[j] = (uint64_t(top_all[j]) * 20ull < uint64_t(m)) ? 1 : 0;
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_copy_prefix_valid(int first, const uint8_t* __restrict__ vin, uint8_t* __restrict__ vout, uint32_t* __restrict__ ctl)
{
    int i = blockIdx.x * IOC_BLOCK + threadIdx.x;
    if (i < first) vout[i] = vin[i];
    if (ctl && i == 0) {  // the sweep's control words: first changed query, the two queue counters, "queue overflowed"
        ctl[0] = 0xFFFFFFFFu;
        ctl[1] = 0u;
        ctl[2] = 0u;
        ctl[3] = 0u;
    }
}


// =====================================================================================================
// launchers
// =====================================================================================================
extern "C" {

hipError_t iock_decide_sweep(hipStream_t st, const void* args_, int nblocks, int eval_blocks, uint32_t* q_count2)
{
    DecideArgs a = *reinterpret_cast<const DecideArgs*>(args_);
    if (a.own_stride > 1) nblocks = owned_count(a.first, a.n, a.own_stride, a.own_offset);  // this rank's queries from a.first on
    if (nblocks <= 0) return hipSuccess;
    a.phase = 1;
    hipLaunchKernelGGL(k_decide_scan, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_eval, dim3(eval_blocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_decide_pick, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    if (a.lazy) return hipGetLastError();
    a.phase = 2;
    a.q_count = q_count2;
    hipLaunchKernelGGL(k_decide_scan, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_eval, dim3(eval_blocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_decide_pick, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    return hipGetLastError();
}

// the second half of an exact sweep alone: for the queries the last lazy sweep left provisional (done == 0), on the cut / top
// that sweep computed
hipError_t iock_decide_phase2(hipStream_t st, const void* args_, int nblocks, int eval_blocks, uint32_t* q_count2)
{
    DecideArgs a = *reinterpret_cast<const DecideArgs*>(args_);
    if (a.own_stride > 1) nblocks = owned_count(a.first, a.n, a.own_stride, a.own_offset);
    if (nblocks <= 0) return hipSuccess;
    a.lazy = 0;
    a.phase = 2;
    a.q_count = q_count2;
    hipLaunchKernelGGL(k_decide_scan, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_eval, dim3(eval_blocks), dim3(IOC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_decide_pick, dim3(nblocks), dim3(IOC_BLOCK), 0, st, a);
    return hipGetLastError();
}

// sharded merge: what this rank does not own is zeroed, so that an all-reduce (maximum of bytes / sum of words) over the ranks
// is a gather by owner; thread 0 also complements the "incomplete" word next to first_changed, so that ONE all-reduce with
// minimum over the control words carries the minimum of first_changed and the maximum of incomplete
__global__ void __launch_bounds__(256) k_shard_mask_u8(uint8_t* __restrict__ a, uint8_t* __restrict__ b, int from, int n, int stride, int offset,
                                                        uint32_t* __restrict__ ctl)
{
    const int j = from + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (blockIdx.x == 0 && threadIdx.x == 0 && ctl) ctl[2] = ~ctl[2];
    if (j >= n) return;
    if ((j % stride) != offset) {
        a[j] = 0;
        if (b) b[j] = 0;
    }
}
__global__ void __launch_bounds__(256) k_shard_mask_i32(int32_t* __restrict__ a, int n, int stride, int offset)
{
    const int j = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j < n && (j % stride) != offset) a[j] = 0;
}
hipError_t iock_shard_mask_u8(hipStream_t st, uint8_t* a, uint8_t* b, int from, int n, int stride, int offset, uint32_t* ctl)
{
    const int m = n - from;
    hipLaunchKernelGGL(k_shard_mask_u8, dim3(m > 0 ? (m + 255) / 256 : 1), dim3(256), 0, st, a, b, from, n, stride, offset, ctl);
    return hipGetLastError();
}
hipError_t iock_shard_mask_i32(hipStream_t st, int32_t* a, int n, int stride, int offset)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_shard_mask_i32, dim3((n + 255) / 256), dim3(256), 0, st, a, n, stride, offset);
    return hipGetLastError();
}

hipError_t iock_gap_bounds(hipStream_t st, int n, const int64_t* off_fwd, const int64_t* off_rev, const uint32_t* pos,
                           const uint32_t* hpc_len, const uint8_t* err_cell, const int32_t* glim, uint2* out, const uint32_t* min_total,
                           uint32_t keep, uint32_t* keep_q)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gap_bounds, dim3(n), dim3(IOC_BLOCK), 0, st, n, off_fwd, off_rev, pos, hpc_len, err_cell, glim, out, min_total, keep,
                       keep_q);
    return hipGetLastError();
}

hipError_t iock_guess_valid(hipStream_t st, int n, const int64_t* off_fwd, const int64_t* off_rev,
                            const uint32_t* top_all, uint8_t* valid)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_guess_valid, dim3((n + IOC_BLOCK - 1) / IOC_BLOCK), dim3(IOC_BLOCK), 0, st, n, off_fwd, off_rev,
                       top_all, valid);
    return hipGetLastError();
}

// the prefix of `valid` that is final already, and (ctl != null) the reset of the sweep's control words in the same launch
hipError_t iock_copy_prefix_valid(hipStream_t st, int first, const uint8_t* vin, uint8_t* vout, uint32_t* ctl)
{
    if (first <= 0 && !ctl) return hipSuccess;
    const int nb = first > 0 ? (first + IOC_BLOCK - 1) / IOC_BLOCK : 1;
    hipLaunchKernelGGL(k_copy_prefix_valid, dim3(nb), dim3(IOC_BLOCK), 0, st, first, vin, vout, ctl);
    return hipGetLastError();
}


size_t iock_decide_args_size() { return sizeof(DecideArgs); }


}  // extern "C"

// (ioc_ctx_prewarm: makes the runtime load this file's code object now instead of at its first launch)
extern "C" hipError_t iock_warm_resolve()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_copy_prefix_valid));
}
