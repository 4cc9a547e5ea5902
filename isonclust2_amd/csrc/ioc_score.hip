// ioc_score.hip — shared-minimizer scoring of the read->cluster assignment path on CDNA4 (gfx950): GetMinimizerHits +
// ConsolidateMinimizerHits + the Size of SortMinimizerHits (src/minimizer.cpp:44-76, src/cluster.cpp:609-636) as LDS
// histograms over the XCD-partitioned index (k_partition_mins, k_score_part, k_score_compact; k_score_t for target ranges
// beyond one histogram), and the full hit table of a query for the tie replay (k_query_table*).  Split out of ioc_kernels.hip
// in round 4; device helpers in ioc_kdev.h.
#include "ioc_kdev.h"

// =====================================================================================================
// k_score — the dominant kernel.  One workgroup per query j (heaviest first).  LDS holds the dense
// histogram Size[strand][target] over the visible targets t < L + j (in passes of `range` targets).
// Each wave takes 64 minimizer occurrences at a time, one per lane:
//   (1) hash probe  -> posting list (off, cnt)                        [64 independent loads in flight]
//   (2) lower_bound -> the part of the ascending list inside the visible window
//   (3) wave prefix sum of the 64 effective lengths, lists compacted into per-wave LDS scratch
//   (4) flattened traversal: lane x of step s owns posting s*64+x of the concatenation, finds its
//       list by a 6-step search over the prefix sums, loads the posting (independent of every other
//       step -> deep memory-level parallelism, all lanes busy) and counts it with an LDS atomic.
// Output: compacted candidate list (target<<1|strandbit, Size) for Size >= keep, ordered by
// (strand, target) — deterministic.
// =====================================================================================================
template <typename PT>
__device__ __forceinline__ uint32_t list_lower_bound(const PT* __restrict__ p, uint32_t n, uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (p[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Flattened traversal of the (compacted) posting lists of one wave chunk, in 16-byte units.
// Every posting list starts 16-byte aligned and is padded to a multiple of 4 entries with
// 0xFFFFFFFF, so a lane fetches 4 postings per load (1 KiB per wave instruction).  The concatenation
// of the nl lists has `total` units; lane x of step s owns unit p = 64*s + x.  Which list p belongs to
// is read off a bitmap of list starts over the concatenation (one 64-bit word per step, built with
// one ds_or per list): list(p) = #starts <= p, a running popcount — two LDS reads per UNIT instead of
// a 6-step search per posting.  The kernel is VALU-issue bound, so instructions per posting are what
// counts: the list bookkeeping is amortised over 4 postings.
#define IOC_BM_WORDS 128  // + IOC_FLAT_UNROLL words of slack are allocated
template <int V, typename PT>
__device__ __forceinline__ void flat_traverse(const PT* __restrict__ post, uint32_t o, uint32_t len,
                                              uint32_t* __restrict__ wb, unsigned long long* __restrict__ bm,
                                              uint32_t* __restrict__ h, uint32_t rbase, uint32_t hi,
                                              unsigned long long& trav, uint32_t& abl)
{
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long le_mask = lt_mask | (1ull << lane);
    constexpr uint32_t PER = 16u / uint32_t(sizeof(PT));  // postings per 16-byte unit: 4 or 8
    constexpr uint32_t PSH = PER == 8 ? 3u : 2u;
    const uint32_t lenU = (len + PER - 1u) >> PSH;
    const unsigned long long nz = __ballot(lenU != 0);
    const uint32_t nl = uint32_t(__popcll(nz));
    if (nl == 0) return;
    const uint32_t incl = wave_incl_scan(lenU);
    const uint32_t total = __shfl(incl, 63);
    const uint32_t excl = incl - lenU;
    const uint32_t nwords = (total + 63) >> 6;
    const uint4* __restrict__ post4 = reinterpret_cast<const uint4*>(post);
    trav += (unsigned long long)PER * total;
    if (nwords <= IOC_BM_WORDS) {
        // (zero IOC_FLAT_UNROLL words past the end so that the unrolled loop reads unconditionally)
        for (uint32_t w = lane; w < nwords + IOC_FLAT_UNROLL; w += 64) bm[w] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lenU) {
            wb[__popcll(nz & lt_mask)] = (o >> PSH) - excl;  // unit address = wb[list] + p
            atomicOr(&bm[excl >> 6], 1ull << (excl & 63u));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t c0 = 0;  // lists started before the current word (same in every lane)
        for (uint32_t w0 = 0; w0 < nwords; w0 += IOC_FLAT_UNROLL) {
            // branch-free body: all LDS reads, then all posting loads, then all LDS atomics are issued
            // back to back
            unsigned long long B[IOC_FLAT_UNROLL];
            uint32_t r[IOC_FLAT_UNROLL], base[IOC_FLAT_UNROLL];
            uint4 tg[IOC_FLAT_UNROLL];
#pragma unroll
            for (int u = 0; u < IOC_FLAT_UNROLL; ++u) B[u] = bm[w0 + u];
#pragma unroll
            for (int u = 0; u < IOC_FLAT_UNROLL; ++u) {
                const uint32_t rr = c0 + uint32_t(__popcll(B[u] & le_mask)) - 1u;
                r[u] = rr < 64u ? rr : 63u;
                c0 += uint32_t(__popcll(B[u]));
            }
#pragma unroll
            for (int u = 0; u < IOC_FLAT_UNROLL; ++u) base[u] = wb[r[u]];
#pragma unroll
            for (int u = 0; u < IOC_FLAT_UNROLL; ++u) {
                const uint32_t p = (w0 + u) * 64u + uint32_t(lane);
                const bool in = p < total;
                const uint32_t a = in ? base[u] + p : 0u;
                if (V == 2 || V == 6 || V == 7) {  // ablation: no posting loads
                    tg[u] = make_uint4(a & 2047u, (a + 1) & 2047u, (a + 2) & 2047u, (a + 3) & 2047u);
                } else {
                    tg[u] = post4[a];
                }
                if (!in) tg[u] = make_uint4(IOC_EMPTY, IOC_EMPTY, IOC_EMPTY, IOC_EMPTY);
            }
#pragma unroll
            for (int u = 0; u < IOC_FLAT_UNROLL; ++u) {
                uint32_t t4[PER];
                if (PER == 4) {
                    t4[0] = tg[u].x;
                    t4[1] = tg[u].y;
                    t4[2] = tg[u].z;
                    t4[3] = tg[u].w;
                } else {
                    const uint32_t w4[4] = {tg[u].x, tg[u].y, tg[u].z, tg[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t4[2 * e] = w4[e] & 0xFFFFu;
                        t4[2 * e + 1] = w4[e] >> 16;
                    }
                }
#pragma unroll
                for (int e = 0; e < int(PER); ++e) {
                    if (V == 1 || V == 6) {  // ablation: no LDS atomics
                        if (t4[e] < hi) abl += t4[e];
                    } else if (V == 7) {  // ablation: plain LDS stores instead of atomics
                        if (t4[e] < hi) h[t4[e] - rbase] = t4[e];
                    } else {
                        // ascending list: entries >= hi (later targets, padding) are not visible;
                        // t - rbase wraps for entries below a range pass's window
                        if (t4[e] - rbase < hi - rbase) atomicAdd(&h[t4[e] - rbase], 1u);
                    }
                }
            }
        }
    } else {
        // very long chunk (> 32768 postings): 6-step search over the prefix sums kept in wb / bm storage
        uint32_t* wx = reinterpret_cast<uint32_t*>(bm);  // 64 words used
        __builtin_amdgcn_wave_barrier();
        if (lenU) {
            const uint32_t r = uint32_t(__popcll(nz & lt_mask));
            wx[r] = excl;
            wb[r] = o >> PSH;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t p = lane; p < total; p += 64) {
            uint32_t r = 0;
#pragma unroll
            for (uint32_t hh = 32; hh > 0; hh >>= 1) {
                const uint32_t r2 = r + hh;
                if (r2 < nl && wx[r2] <= p) r = r2;
            }
            const uint4 t = post4[wb[r] + (p - wx[r])];
            const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (PER == 4) {
                    if (w4[e] - rbase < hi - rbase) atomicAdd(&h[w4[e] - rbase], 1u);
                } else {
                    const uint32_t a0 = w4[e] & 0xFFFFu, a1 = w4[e] >> 16;
                    if (a0 - rbase < hi - rbase) atomicAdd(&h[a0 - rbase], 1u);
                    if (a1 - rbase < hi - rbase) atomicAdd(&h[a1 - rbase], 1u);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- u16 postings, single pass: the per-posting work of flat_traverse cut to 2 VALU ----------------------------------------
// rocprof on round 1's kernel: VALU-issue (68 % busy, 9 VALU per posting slot) and LDS (63 % busy) bound together.  Per
// posting the compiler emitted extract (and / shift) + compare + shift-add + the exec save / restore around the atomic.
// Here the window test runs on the raw 16-bit half of the loaded word (v_cmp_lt_u32_sdwa), the counter's LDS address
// is half * 4 + base in one v_mad_u32_u16 (op_sel picks the half), and the atomic is issued under the compare's mask:
// 2 VALU + 1 ds_add_u32 per posting; padding (0xFFFF) and the entries of the epoch slack fail the test as before.
//
// IOC_SCORE_OOB (default): no window test at all.  k_score_part puts the histogram of its T visible targets at the very END
// of the workgroup's LDS allocation (counter of target t at end - 4 (T - t)), so the counter address of every entry the test
// would reject — targets >= T of the epoch slack, the 0xFFFF padding — lies beyond the allocation, and gfx950 discards an
// LDS atomic there (tools/micro/lds_oob.hip, profiles/r02_lds_oob.txt: the hardware's bound is the allocation rounded up
// to its 1280-byte granule; 1.4·10^11 atomics above it changed no word of any workgroup's memory).  1 VALU + 1 ds_add_u32
// per posting, no VCC / EXEC traffic; lanes past the end of the concatenation get a base far outside instead of T = 0.
template <bool OOB>
__device__ __forceinline__ void count_word_u16(uint32_t w, uint32_t T, uint32_t hbase, uint32_t one)
{
    uint32_t a;
    unsigned long long sv;
#if IOC_SCORE_ABL == 1   // ablation build: no LDS atomics
    asm volatile("v_cmp_lt_u32_sdwa vcc, %2, %3 src0_sel:WORD_0 src1_sel:DWORD\n\tv_mad_u32_u16 %0, %2, 4, %4 op_sel:[0,0,0,0]\n\t"
                 "v_cmp_lt_u32_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:DWORD\n\tv_mad_u32_u16 %0, %2, 4, %4 op_sel:[1,0,0,0]"
                 : "=&v"(a), "=&s"(sv) : "v"(w), "v"(T), "v"(hbase), "v"(one) : "vcc", "memory");
    return;
#elif IOC_SCORE_ABL == 2  // ablation build: conflict-free atomics (every lane its own bank)
    hbase += (threadIdx.x & 31u) * 4u;
    w = 0;
    T = T ? 1u : 0u;
#endif
    if constexpr (OOB && IOC_SCORE_ABL == 0) {
        uint32_t a2;
        (void)sv;
        (void)T;
        asm volatile(
            "v_mad_u32_u16 %0, %2, 4, %3 op_sel:[0,0,0,0]\n\t"
            "v_mad_u32_u16 %1, %2, 4, %3 op_sel:[1,0,0,0]\n\t"
            "ds_add_u32 %0, %4\n\t"
            "ds_add_u32 %1, %4"
            : "=&v"(a), "=&v"(a2)
            : "v"(w), "v"(hbase), "v"(one)
            : "memory");
    } else {
        asm volatile(
            "v_cmp_lt_u32_sdwa vcc, %2, %3 src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "s_and_saveexec_b64 %1, vcc\n\t"
            "v_mad_u32_u16 %0, %2, 4, %4 op_sel:[0,0,0,0]\n\t"
            "ds_add_u32 %0, %5\n\t"
            "s_mov_b64 exec, %1\n\t"
            "v_cmp_lt_u32_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "s_and_saveexec_b64 %1, vcc\n\t"
            "v_mad_u32_u16 %0, %2, 4, %4 op_sel:[1,0,0,0]\n\t"
            "ds_add_u32 %0, %5\n\t"
            "s_mov_b64 exec, %1"
            : "=&v"(a), "=&s"(sv)
            : "v"(w), "v"(T), "v"(hbase), "v"(one)
            : "vcc", "memory");
    }
}

// (tuning, round 4 — profiles/r04_score_variants.txt: groups of 4 steps instead of 8, a tail of single steps, the next chunk's
// hash probes in flight while the current chunk is traversed and a register budget of 64 (8 waves per SIMD, a handful of
// spilled registers) took the scoring phase of config 2 from 0.774 to 0.696 ms: the kernel waits more than it issues, and what
// it waits for — index rows, posting units, the LDS atomic pipe — is covered by more resident waves, not by a longer group)
#ifndef IOC_FLAT_UNROLL16
#define IOC_FLAT_UNROLL16 4
#endif
#ifndef IOC_FLAT_TAIL16
#define IOC_FLAT_TAIL16 1
#endif
static_assert(IOC_FLAT_UNROLL16 <= IOC_FLAT_UNROLL, "the bitmap's slack words are sized by IOC_FLAT_UNROLL");
template <bool OOB>
__device__ __forceinline__ void flat_traverse_u16(const uint16_t* __restrict__ post, uint32_t o, uint32_t len,
                                                  uint32_t* __restrict__ wb, unsigned long long* __restrict__ bm,
                                                  uint32_t* __restrict__ h, uint32_t T, unsigned long long& trav, uint32_t& abl)
{
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long le_mask = lt_mask | (1ull << lane);
    const uint32_t lenU = (len + 7u) >> 3;
    const unsigned long long nz = __ballot(lenU != 0);
    if (nz == 0ull) return;
    const uint32_t incl = wave_incl_scan(lenU);
    const uint32_t total = __shfl(incl, 63);
    const uint32_t excl = incl - lenU;
    const uint32_t nwords = (total + 63) >> 6;
    if (nwords > IOC_BM_WORDS) {  // a very long chunk: the general path (6-step search)
        flat_traverse<0, uint16_t>(post, o, len, wb, bm, h, 0u, T, trav, abl);
        return;
    }
    const uint4* __restrict__ post4 = reinterpret_cast<const uint4*>(post);
    trav += IOC_SCORE_TRAV_CAPACITY ? 512ull * (nwords / IOC_FLAT_UNROLL16 * IOC_FLAT_UNROLL16 + (nwords % IOC_FLAT_UNROLL16 + IOC_FLAT_TAIL16 - 1) / IOC_FLAT_TAIL16 * IOC_FLAT_TAIL16) : 8ull * total;
    for (uint32_t w = lane; w < nwords + IOC_FLAT_UNROLL16; w += 64) bm[w] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lenU) {
        wb[__popcll(nz & lt_mask)] = (o >> 3) - excl;  // unit address = wb[list] + p
        atomicOr(&bm[excl >> 6], 1ull << (excl & 63u));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t hbase = uint32_t(reinterpret_cast<uintptr_t>(h));  // LDS byte address of the strand's histogram
    const uint32_t one = 1u;
    uint32_t c0 = 0;
    // U steps of 64 units at a time: lookups, loads and counting of the U steps are interleaved by the compiler
    auto group = [&](auto ucount, uint32_t w0) {
        constexpr int U = decltype(ucount)::value;
        unsigned long long B[U];
        uint32_t r[U], base[U];
        uint4 tg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) B[u] = bm[w0 + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t rr = c0 + uint32_t(__popcll(B[u] & le_mask)) - 1u;
            r[u] = rr < 64u ? rr : 63u;
            c0 += uint32_t(__popcll(B[u]));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) base[u] = wb[r[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = (w0 + u) * 64u + uint32_t(lane);
            tg[u] = post4[p < total ? base[u] + p : 0u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (a lane past the end of the concatenation loaded unit 0: a window of 0 targets rejects all of it)
            const bool inl = (w0 + u) * 64u + uint32_t(lane) < total;
            const uint32_t Tl = inl ? T : 0u;
            const uint32_t hb = (OOB && IOC_SCORE_ABL == 0) ? (inl ? hbase : IOC_OOB_FAR_BASE) : hbase;  // (1 MB: outside any LDS)
            count_word_u16<OOB>(tg[u].x, Tl, hb, one);
            count_word_u16<OOB>(tg[u].y, Tl, hb, one);
            count_word_u16<OOB>(tg[u].z, Tl, hb, one);
            count_word_u16<OOB>(tg[u].w, Tl, hb, one);
        }
    };
    // whole groups of IOC_FLAT_UNROLL16 steps, then the rest two steps at a time: with one loop of 8 the steps past the end of
    // a chunk (9.8 steps on average on config 2) were 31 % of all the posting slots the kernel issued
    uint32_t w0 = 0;
    for (; w0 + IOC_FLAT_UNROLL16 <= nwords; w0 += IOC_FLAT_UNROLL16) group(std::integral_constant<int, IOC_FLAT_UNROLL16>{}, w0);
    for (; w0 < nwords; w0 += IOC_FLAT_TAIL16) group(std::integral_constant<int, IOC_FLAT_TAIL16>{}, w0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the atomics above are invisible to the compiler's counters)
    __builtin_amdgcn_wave_barrier();
}

template <int V, typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_score_t(int n, uint32_t L, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
        const uint32_t* __restrict__ mins, const uint4* __restrict__ rows, uint32_t cap, uint32_t shift,
        const PT* __restrict__ post, uint32_t range, uint32_t keep, uint32_t* __restrict__ cand_key,
        uint32_t* __restrict__ cand_size, uint32_t* __restrict__ cand_count,
        unsigned long long* __restrict__ traversed, Epochs E,
        const uint8_t* __restrict__ audit_valid, unsigned long long* __restrict__ audit_sum, int own_stride, int own_offset,
        const uint32_t* __restrict__ keep_q)
{
    extern __shared__ uint32_t hist[];  // 2 * min(range, L + j)
    __shared__ uint32_t wcount[IOC_WAVES];
    __shared__ uint32_t s_wb[IOC_WAVES][64];                       // per compacted list: address base
    __shared__ unsigned long long s_bm[IOC_WAVES][IOC_BM_WORDS + IOC_FLAT_UNROLL];   // bitmap of list starts
    const int j = owned_from_top(n, int(blockIdx.x), own_stride, own_offset);
    if (j < 0) return;
    if (keep_q) keep = keep_q[j];  // (fast mode: below this Size no candidate of this query can pass, see k_gap_bounds)
    const uint32_t T = L + uint32_t(j);  // visible targets: [0, T)
    // first epoch boundary >= T: the field of the row info that holds its cut
    uint32_t eword, eshift;
    epoch_field(E, T, eword, eshift);
    const int lane = lane_id(), wave = wave_id();
    const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
    uint32_t written = 0;
    unsigned long long trav = 0;
    uint32_t abl = 0;
    uint32_t* const wb_ = s_wb[wave];
    unsigned long long* const bm_ = s_bm[wave];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (uint32_t rbase = 0; rbase < T; rbase += range) {
        const uint32_t Tr = (T - rbase < range) ? (T - rbase) : range;
        for (uint32_t i = threadIdx.x; i < 2 * Tr; i += IOC_BLOCK) hist[i] = 0;
        __syncthreads();
        const uint32_t hi = rbase + Tr;  // count targets in [rbase, hi)
        for (int s = 0; s < 2; ++s) {
            const int64_t b = s == 0 ? off_fwd[j] : off_rev[j];
            const int64_t e = s == 0 ? off_fwd[j + 1] : off_rev[j + 1];
            uint32_t* h = hist + uint32_t(s) * Tr;
            // software pipelining: the hash probe of the next chunk is issued before this chunk's
            // postings are traversed
            uint32_t o_nx = 0, c_nx = 0;
            uint2 q_nx = make_uint2(0u, 0u);
            {
                const int64_t t = b + wave * 64 + lane;
                if (t < e) index_lookup(rows, cap, shift, mins[t], o_nx, c_nx, q_nx);
            }
            if (V == 5) continue;  // ablation: no probes, no traversal
            for (int64_t c0 = b + wave * 64; c0 < e; c0 += IOC_WAVES * 64) {
                uint32_t o = o_nx, len = c_nx;
                const uint2 qi = q_nx;
                {
                    const int64_t t = c0 + IOC_WAVES * 64 + lane;
                    o_nx = 0;
                    c_nx = 0;
                    q_nx = make_uint2(0u, 0u);
                    if (t < e) index_lookup(rows, cap, shift, mins[t], o_nx, c_nx, q_nx);
                }
                // Visible part of the ascending list.  Single pass (the common case): the row carries
                // the list positions of three epoch boundaries, so the list is cut at the first
                // boundary >= T without touching it; the few entries in [T, boundary) are rejected by
                // the window test below.  Long lists and range passes pay a binary search.
                if (len) {
                    if (rbase == 0 && hi == T && !(qi.y & 0x80000000u)) {
                        len = epoch_cut(qi, len, eword, eshift);
                    } else {
                        const PT* pl = post + o;
                        // (start rounded down to a 16-byte unit; entries < rbase are rejected below)
                        const uint32_t i0 = rbase ? (list_lower_bound(pl, len, rbase) & ~(16u / uint32_t(sizeof(PT)) - 1u)) : 0u;
                        const uint32_t i1 = list_lower_bound(pl, len, hi);
                        len = i1 - i0;
                        o += i0;
                    }
                }
                if (V == 4) {  // ablation: probes only
                    abl += len + o;
                } else {
                    flat_traverse<V, PT>(post, o, len, wb_, bm_, h, rbase, hi, trav, abl);
                }
            }
        }
        __syncthreads();
        if (audit_valid) {
            // instrumentation launch: number of postings the reference would traverse for this query =
            // sum of Size over the targets that are clusters (GetMinimizerHits raw hits)
            unsigned long long sum = 0;
            for (uint32_t i = threadIdx.x; i < 2 * Tr; i += IOC_BLOCK) {
                const uint32_t tg = rbase + (i >= Tr ? i - Tr : i);
                if (tg < L || audit_valid[tg - L]) sum += hist[i];
            }
            for (int o2 = 32; o2 > 0; o2 >>= 1) sum += __shfl_down(sum, o2);
            if (lane == 0 && sum) atomicAdd(audit_sum, sum);
            __syncthreads();
            continue;
        }
        // ---- ordered compaction of hist[0 .. 2*Tr) --------------------------------------------
        const uint32_t tot = 2 * Tr;
        const uint32_t per = (tot + IOC_WAVES - 1) / IOC_WAVES;
        const uint32_t w0 = wave * per;
        const uint32_t w1 = (w0 + per < tot) ? (w0 + per) : tot;
        uint32_t my = 0;
        for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
            uint32_t i = i0 + lane;
            bool f = (i < w1) && (hist[i] >= keep);
            my += __popcll(__ballot(f));
        }
        if (lane == 0) wcount[wave] = my;
        __syncthreads();
        uint32_t wbase = written, all = 0;
        for (int w = 0; w < IOC_WAVES; ++w) {
            if (w < wave) wbase += wcount[w];
            all += wcount[w];
        }
        for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
            uint32_t i = i0 + lane;
            uint32_t v = (i < w1) ? hist[i] : 0;
            bool f = (i < w1) && (v >= keep);
            unsigned long long bm = __ballot(f);
            if (f) {
                uint32_t pos = wbase + __popcll(bm & lt_mask);
                uint32_t strandbit = (i >= Tr) ? 1u : 0u;
                uint32_t tg = rbase + (strandbit ? i - Tr : i);
                cand_key[cbase + pos] = (tg << 1) | strandbit;
                cand_size[cbase + pos] = v;
            }
            wbase += __popcll(bm);
        }
        written += all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && !audit_valid) cand_count[j] = written;
    if (V != 0 && abl == 0x12345678u) cand_count[j] = abl;  // keeps the ablated loads alive
    if (traversed && lane == 0) atomicAdd(traversed, trav);
}

// =====================================================================================================
// XCD-partitioned scoring (single-pass case).  The index (rows + postings) is several times larger
// than one XCD's 4 MiB L2, and a query's probes are random, so the plain kernel misses L2 on >80 % of
// its requests.  Here the value space is cut into 8 partitions by the top 3 bits of the hash slot —
// rows and postings of a partition are contiguous — and workgroup (query j, partition x) has
// blockIdx = 8*j' + x.  Workgroups are dealt round-robin over the 8 XCDs, so partition x is only
// ever touched from one XCD and its slice of the index stays L2-resident (placement is a speed
// assumption only: any mapping gives the same result).  Each workgroup keeps a private LDS histogram
// of its partition's hits and stores it; k_score_compact adds the 8 partial histograms of a query
// and writes the candidate list.
// =====================================================================================================
#define IOC_PARTS 8
// Minimizer values of every query, bucketed by index partition (order inside a bucket is irrelevant to a
// histogram): pmins holds a permutation of mins per (query, strand), pbnd the 9 bucket boundaries.
__global__ void __launch_bounds__(IOC_BLOCK)
k_partition_mins(int n, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
                 const uint32_t* __restrict__ mins, uint32_t shift, uint32_t* __restrict__ pmins,
                 uint32_t* __restrict__ pbnd, uint32_t* __restrict__ max_len, int own_stride, int own_offset)
{
    __shared__ uint32_t cnt[IOC_PARTS], cur[IOC_PARTS];
    const int j = owned_from(0, int(blockIdx.x), own_stride, own_offset);
    if (j >= n) return;
    const uint32_t pshift = (32u - shift) - 3u;
    for (int s = 0; s < 2; ++s) {
        const int64_t b = s == 0 ? off_fwd[j] : off_rev[j];
        const int64_t e = s == 0 ? off_fwd[j + 1] : off_rev[j + 1];
        if (threadIdx.x < IOC_PARTS) cnt[threadIdx.x] = 0;
        // a Size can never exceed the strand's minimizer count: below 65536 the partial histograms are u16
        if (threadIdx.x == 0 && uint32_t(e - b) > 65535u) atomicMax(max_len, uint32_t(e - b));
        __syncthreads();
        for (int64_t t = b + threadIdx.x; t < e; t += IOC_BLOCK) {
            const uint32_t v = mins[t];
            atomicAdd(&cnt[(v == IOC_EMPTY) ? 0u : (hash_slot(v, shift) >> pshift)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            uint32_t* out = pbnd + (size_t(j) * 2 + s) * (IOC_PARTS + 1);
            for (int x = 0; x < IOC_PARTS; ++x) {
                out[x] = acc;
                cur[x] = acc;
                acc += cnt[x];
            }
            out[IOC_PARTS] = acc;
        }
        __syncthreads();
        for (int64_t t = b + threadIdx.x; t < e; t += IOC_BLOCK) {
            const uint32_t v = mins[t];
            const uint32_t pos = atomicAdd(&cur[(v == IOC_EMPTY) ? 0u : (hash_slot(v, shift) >> pshift)], 1u);
            pmins[b + pos] = v;
        }
        __syncthreads();
    }
}

#ifndef IOC_SCORE_PART_MINWAVES
#define IOC_SCORE_PART_MINWAVES 8  // minimum waves per SIMD the register allocation must allow: 64 registers
#endif
#ifndef IOC_SCORE_PREFETCH
#define IOC_SCORE_PREFETCH 1       // 1: the hash probe of a wave's NEXT chunk of minimizers is issued before the current chunk is traversed
#endif
template <typename PT, bool OOB>
__global__ void __launch_bounds__(IOC_BLOCK, IOC_SCORE_PART_MINWAVES)
k_score_part(int n, uint32_t L, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
             const uint32_t* __restrict__ pmins, const uint32_t* __restrict__ pbnd, const uint4* __restrict__ rows,
             uint32_t cap, uint32_t shift, const PT* __restrict__ post, uint32_t* __restrict__ part, Epochs E,
             unsigned long long* __restrict__ traversed, const uint32_t* __restrict__ max_len, uint32_t dyn_bytes, int own_stride, int own_offset)
{
    // ONE strand's histogram at a time (L + j counters): half the LDS of a both-strands histogram, twice the workgroups per
    // CU (the kernel is bound by latency as much as by VALU issue and LDS conflicts: 17.6 waves per CU with 24 KB per
    // workgroup); a strand's counts go out as soon as it is done
    extern __shared__ uint32_t hist_dyn[];  // >= L + j counters
    __shared__ uint32_t s_wb[IOC_WAVES][64];
    __shared__ unsigned long long s_bm[IOC_WAVES][IOC_BM_WORDS + IOC_FLAT_UNROLL];
    const int j = owned_from_top(n, int(blockIdx.x / IOC_PARTS), own_stride, own_offset);
    const uint32_t x = blockIdx.x % IOC_PARTS;
    if (j < 0) return;
    const uint32_t T = L + uint32_t(j);
    if (T == 0) return;
    // the T counters end where the workgroup's LDS allocation ends for the hardware (dynamic memory is the last thing in
    // it; the allocation is a whole number of 1280-byte granules on gfx950): see count_word_u16
    const uint32_t dyn_base = uint32_t(reinterpret_cast<uintptr_t>(hist_dyn));
    const uint32_t lds_end = (dyn_base + dyn_bytes + 1279u) / 1280u * 1280u;
    uint32_t* const hist = OOB ? hist_dyn + ((lds_end - dyn_base) / 4u - T) : hist_dyn;
    uint32_t eword, eshift;
    epoch_field(E, T, eword, eshift);
    const int lane = lane_id(), wave = wave_id();
    const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
    uint32_t* const wb_ = s_wb[wave];
    unsigned long long* const bm_ = s_bm[wave];
    unsigned long long trav = 0;
    uint32_t abl = 0;
    const bool narrow = *max_len < 65536u;  // two u16 counts per word (cbase and 2T are even)
    for (int s = 0; s < 2; ++s) {
        for (uint32_t i = threadIdx.x; i < T; i += IOC_BLOCK) hist[i] = 0;
        __syncthreads();
        const int64_t b0 = s == 0 ? off_fwd[j] : off_rev[j];
        const uint32_t* bnd = pbnd + (size_t(j) * 2 + s) * (IOC_PARTS + 1);
        const int64_t b = b0 + bnd[x], e = b0 + bnd[x + 1];  // this partition's bucket
        uint32_t* h = hist;
#if IOC_SCORE_PREFETCH
        uint32_t o_nx = 0, c_nx = 0;
        uint2 q_nx = make_uint2(0u, 0u);
        {
            const int64_t t = b + wave * 64 + lane;
            if (t < e) index_lookup(rows, cap, shift, pmins[t], o_nx, c_nx, q_nx);
        }
#endif
        for (int64_t c0 = b + wave * 64; c0 < e; c0 += IOC_WAVES * 64) {
#if IOC_SCORE_PREFETCH
            uint32_t o = o_nx, len = c_nx;
            const uint2 qi = q_nx;
            {
                const int64_t t = c0 + IOC_WAVES * 64 + lane;
                o_nx = 0;
                c_nx = 0;
                q_nx = make_uint2(0u, 0u);
                if (t < e) index_lookup(rows, cap, shift, pmins[t], o_nx, c_nx, q_nx);
            }
#else
            const int64_t t = c0 + lane;
            uint32_t o = 0, len = 0;
            uint2 qi = make_uint2(0u, 0u);
            if (t < e) index_lookup(rows, cap, shift, pmins[t], o, len, qi);
#endif
            if (len) {
                if (!(qi.y & 0x80000000u))
                    len = epoch_cut(qi, len, eword, eshift);
                else
                    len = list_lower_bound(post + o, len, T);
            }
            if (sizeof(PT) == 2 && !IOC_SCORE_OLD_TRAVERSE)
                flat_traverse_u16<OOB>(reinterpret_cast<const uint16_t*>(post), o, len, wb_, bm_, h, T, trav, abl);
            else
                flat_traverse<0, PT>(post, o, len, wb_, bm_, h, 0u, T, trav, abl);
        }
        __syncthreads();
        // the partial histogram of (query, partition) is [strand][target]: this strand's slice
        if (narrow) {
            uint16_t* out = reinterpret_cast<uint16_t*>(part + (IOC_PARTS * cbase) / 2 + size_t(x) * T) + size_t(s) * T;
            for (uint32_t i = threadIdx.x; i < T; i += IOC_BLOCK) out[i] = uint16_t(hist[i]);
        } else {
            uint32_t* out = part + IOC_PARTS * cbase + size_t(x) * 2 * T + size_t(s) * T;
            for (uint32_t i = threadIdx.x; i < T; i += IOC_BLOCK) out[i] = hist[i];
        }
        __syncthreads();
    }
    if (traversed && lane == 0) atomicAdd(traversed, trav);
}

// ---- run-time check of what the OOB variant of k_score_part relies on -------------------------------------------------------
// k_score_part<PT, true> has no window test: the counter address of every posting the test would reject lies in
// [lds_end, lds_end + 256 KB) or in [IOC_OOB_FAR_BASE, IOC_OOB_FAR_BASE + 256 KB), where lds_end is the workgroup's LDS
// allocation (static + dynamic) rounded up to the hardware's 1280-byte granule, and the variant is right iff the hardware
// drops an LDS atomic there.  That is gfx950 behaviour, not a documented guarantee, so ioc_ctx_create PROBES it on the device
// it runs on, with k_score_part's own static LDS layout and two dynamic sizes: every workgroup of a grid that fills the chip
// several times over (so that workgroups share CUs) paints its whole allocation, issues one atomic to EVERY word of both
// ranges, checks that a counter in the granule's slack still counts (the histogram lives there) and that no word of its
// allocation changed — its own stray atomics would show, and so would a neighbour's.  result[0]: bit 0 = a word changed,
// bit 1 = an in-bounds atomic was lost; result[1] = workgroups that ran.  A failed probe selects the masked variant.
__global__ void __launch_bounds__(IOC_BLOCK) k_lds_oob_probe(uint32_t dyn_bytes, uint32_t* __restrict__ result)
{
    extern __shared__ uint32_t hist_dyn[];
    __shared__ uint32_t s_wb[IOC_WAVES][64];
    __shared__ unsigned long long s_bm[IOC_WAVES][IOC_BM_WORDS + IOC_FLAT_UNROLL];
    s_wb[0][threadIdx.x & 63] = 0;  // (keeps the static arrays, and with them k_score_part's dynamic base, in the kernel)
    s_bm[0][0] = 0ull;
    const uint32_t dyn_base = uint32_t(reinterpret_cast<uintptr_t>(hist_dyn));
    const uint32_t lds_end = (dyn_base + dyn_bytes + 1279u) / 1280u * 1280u;
    const uint32_t salt = 0x9E3779B9u * (blockIdx.x + 1u);
    __syncthreads();
    for (uint32_t a = threadIdx.x * 4u; a < lds_end; a += IOC_BLOCK * 4u) {
        const uint32_t v = a * 2654435761u ^ salt;
        asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t one = 1u;
    // in bounds, in the slack behind the requested dynamic size: the last IOC_BLOCK words of the allocation
    const uint32_t in_a = lds_end - 4u * (threadIdx.x + 1u);
    asm volatile("ds_add_u32 %0, %1" ::"v"(in_a), "v"(one) : "memory");
    for (uint32_t r = 0; r < 2; ++r) {
        const uint32_t base = r == 0 ? lds_end : IOC_OOB_FAR_BASE;
        for (uint32_t a = threadIdx.x * 4u; a < 0x40000u + 1280u; a += IOC_BLOCK * 4u) {
            const uint32_t t = base + a;
            asm volatile("ds_add_u32 %0, %1" ::"v"(t), "v"(one) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t bad = 0;
    for (uint32_t a = threadIdx.x * 4u; a < lds_end; a += IOC_BLOCK * 4u) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        const uint32_t want = (a * 2654435761u ^ salt) + (a + 4u * IOC_BLOCK >= lds_end ? 1u : 0u);
        if (v != want) bad |= (a + 4u * IOC_BLOCK >= lds_end && v == want - 1u) ? 2u : 1u;
    }
    if (bad) atomicOr(&result[0], bad);
    if (threadIdx.x == 0) atomicAdd(&result[1], 1u);
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_score_compact(int n, uint32_t L, const uint32_t* __restrict__ part, uint32_t keep,
                uint32_t* __restrict__ cand_key, uint32_t* __restrict__ cand_size,
                uint32_t* __restrict__ cand_count, const uint8_t* __restrict__ audit_valid,
                unsigned long long* __restrict__ audit_sum, uint32_t* __restrict__ top_all,
                const uint32_t* __restrict__ max_len, int own_stride, int own_offset, const uint32_t* __restrict__ keep_q)
{
    extern __shared__ uint32_t hist[];  // 2 * (L + j)
    __shared__ uint32_t wcount[IOC_WAVES];
    __shared__ uint32_t wtop[IOC_WAVES];
    const int j = owned_from_top(n, int(blockIdx.x), own_stride, own_offset);
    if (j < 0) return;
    if (keep_q) keep = keep_q[j];
    const uint32_t T = L + uint32_t(j);
    const int lane = lane_id(), wave = wave_id();
    const uint64_t cbase = 2ull * L * uint64_t(j) + uint64_t(j) * uint64_t(j > 0 ? j - 1 : 0);
    const uint32_t tot = 2 * T;
    uint32_t tmax = 0;
    if (*max_len < 65536u) {
        const uint32_t* src = part + (IOC_PARTS * cbase) / 2;
        for (uint32_t i = threadIdx.x; i < T; i += IOC_BLOCK) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int x = 0; x < IOC_PARTS; ++x) {
                const uint32_t w = src[size_t(x) * T + i];
                lo += w & 0xFFFFu;
                hi += w >> 16;
            }
            hist[2 * i] = lo;
            hist[2 * i + 1] = hi;
            tmax = lo > tmax ? lo : tmax;
            tmax = hi > tmax ? hi : tmax;
        }
    } else {
        const uint32_t* src = part + IOC_PARTS * cbase;
        for (uint32_t i = threadIdx.x; i < tot; i += IOC_BLOCK) {
            uint32_t v = 0;
#pragma unroll
            for (int x = 0; x < IOC_PARTS; ++x) v += src[size_t(x) * tot + i];
            hist[i] = v;
            tmax = v > tmax ? v : tmax;
        }
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const uint32_t t = __shfl_down(tmax, o2);
        tmax = t > tmax ? t : tmax;
    }
    if (lane == 0) wtop[wave] = tmax;
    __syncthreads();
    if (threadIdx.x == 0 && top_all && !audit_valid) {
        uint32_t t = 0;
        for (int w = 0; w < IOC_WAVES; ++w) t = wtop[w] > t ? wtop[w] : t;
        top_all[j] = t;  // largest Size against ANY earlier entry: seeds the resolve's first guess
    }
    if (audit_valid) {
        unsigned long long sum = 0;
        for (uint32_t i = threadIdx.x; i < tot; i += IOC_BLOCK) {
            const uint32_t tg = i >= T ? i - T : i;
            if (tg < L || audit_valid[tg - L]) sum += hist[i];
        }
        for (int o2 = 32; o2 > 0; o2 >>= 1) sum += __shfl_down(sum, o2);
        if (lane == 0 && sum) atomicAdd(audit_sum, sum);
        return;
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t per = (tot + IOC_WAVES - 1) / IOC_WAVES;
    const uint32_t w0 = wave * per;
    const uint32_t w1 = (w0 + per < tot) ? (w0 + per) : tot;
    uint32_t my = 0;
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool f = (i < w1) && (hist[i] >= keep);
        my += __popcll(__ballot(f));
    }
    if (lane == 0) wcount[wave] = my;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
    for (int w = 0; w < IOC_WAVES; ++w) {
        if (w < wave) wbase += wcount[w];
        all += wcount[w];
    }
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t v = (i < w1) ? hist[i] : 0;
        const bool f = (i < w1) && (v >= keep);
        const unsigned long long bm = __ballot(f);
        if (f) {
            const uint32_t pos = wbase + __popcll(bm & lt_mask);
            const uint32_t strandbit = (i >= T) ? 1u : 0u;
            const uint32_t tg = strandbit ? i - T : i;
            cand_key[cbase + pos] = (tg << 1) | strandbit;
            cand_size[cbase + pos] = v;
        }
        wbase += __popcll(bm);
    }
    if (threadIdx.x == 0) cand_count[j] = all;
}


// =====================================================================================================
// k_query_table — full hit table of ONE query against the targets that are clusters (tie replay on
// the host): Size and the Index of the first hitting read minimizer per (target, strand).
// hist/first live in global scratch (2 * T words each), zeroed / set to 0xFFFFFFFF by the host.
// =====================================================================================================
template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_query_table(int j, uint32_t L, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
              const uint32_t* __restrict__ mins, const uint4* __restrict__ rows, uint32_t cap, uint32_t shift,
              const PT* __restrict__ post, const uint8_t* __restrict__ valid, uint32_t* __restrict__ hist,
              uint32_t* __restrict__ first)
{
    const uint32_t T = L + uint32_t(j);
    const int lane = lane_id();
    const uint32_t gw = (blockIdx.x * IOC_BLOCK + threadIdx.x) >> 6;
    const uint32_t nw = (gridDim.x * IOC_BLOCK) >> 6;
    for (int s = 0; s < 2; ++s) {
        const int64_t b = s == 0 ? off_fwd[j] : off_rev[j];
        const int64_t e = s == 0 ? off_fwd[j + 1] : off_rev[j + 1];
        for (int64_t c0 = b + int64_t(gw) * 64; c0 < e; c0 += int64_t(nw) * 64) {
            int64_t t = c0 + lane;
            uint32_t o = 0, c = 0;
            uint2 qi_ = make_uint2(0u, 0u);
            if (t < e) index_lookup(rows, cap, shift, mins[t], o, c, qi_);
            unsigned long long mask = __ballot(c != 0);
            while (mask) {
                int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                uint32_t lo = __builtin_amdgcn_readlane(o, l);
                uint32_t lc = __builtin_amdgcn_readlane(c, l);
                uint32_t idx = uint32_t(c0 + l - b);
                for (uint32_t p = lane; p < lc; p += 64) {
                    uint32_t tg = post[lo + p];
                    if (tg < T && (tg < L || valid[tg - L])) {
                        atomicAdd(&hist[uint32_t(s) * T + tg], 1u);
                        atomicMin(&first[uint32_t(s) * T + tg], idx);
                    }
                }
            }
        }
    }
}

template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_query_table_many(const int32_t* __restrict__ qlist, uint64_t stride, uint32_t L, const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
              const uint32_t* __restrict__ mins, const uint4* __restrict__ rows, uint32_t cap, uint32_t shift,
              const PT* __restrict__ post, const uint8_t* __restrict__ valid, uint32_t* __restrict__ hist_base,
              uint32_t* __restrict__ first_base)
{
    // blockIdx.y = position in the query list; every query has a slice of `stride` words in the two tables
    const int j = qlist[blockIdx.y];
    uint32_t* __restrict__ hist = hist_base + uint64_t(blockIdx.y) * stride;
    uint32_t* __restrict__ first = first_base + uint64_t(blockIdx.y) * stride;
    const uint32_t T = L + uint32_t(j);
    const int lane = lane_id();
    const uint32_t gw = (blockIdx.x * IOC_BLOCK + threadIdx.x) >> 6;
    const uint32_t nw = (gridDim.x * IOC_BLOCK) >> 6;
    for (int s = 0; s < 2; ++s) {
        const int64_t b = s == 0 ? off_fwd[j] : off_rev[j];
        const int64_t e = s == 0 ? off_fwd[j + 1] : off_rev[j + 1];
        for (int64_t c0 = b + int64_t(gw) * 64; c0 < e; c0 += int64_t(nw) * 64) {
            int64_t t = c0 + lane;
            uint32_t o = 0, c = 0;
            uint2 qi_ = make_uint2(0u, 0u);
            if (t < e) index_lookup(rows, cap, shift, mins[t], o, c, qi_);
            unsigned long long mask = __ballot(c != 0);
            while (mask) {
                int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                uint32_t lo = __builtin_amdgcn_readlane(o, l);
                uint32_t lc = __builtin_amdgcn_readlane(c, l);
                uint32_t idx = uint32_t(c0 + l - b);
                for (uint32_t p = lane; p < lc; p += 64) {
                    uint32_t tg = post[lo + p];
                    if (tg < T && (tg < L || valid[tg - L])) {
                        atomicAdd(&hist[uint32_t(s) * T + tg], 1u);
                        atomicMin(&first[uint32_t(s) * T + tg], idx);
                    }
                }
            }
        }
    }
}

// the non-empty cells of every query's table: out slice = [count][idx, Size, first] * cap
__global__ void __launch_bounds__(256) k_query_compact_many(const int32_t* __restrict__ qlist, uint64_t stride, uint32_t L,
                                                             const uint32_t* __restrict__ hist_base, const uint32_t* __restrict__ first_base,
                                                             uint32_t cap, uint32_t* __restrict__ out_base)
{
    const uint32_t n2 = 2u * (L + uint32_t(qlist[blockIdx.y]));
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const uint32_t* hist = hist_base + uint64_t(blockIdx.y) * stride;
    const uint32_t sz = hist[i];
    if (!sz) return;
    uint32_t* out = out_base + uint64_t(blockIdx.y) * (1u + 3u * uint64_t(cap));
    const uint32_t pos = atomicAdd(&out[0], 1u);
    if (pos < cap) {
        out[1 + 3 * pos] = i;
        out[2 + 3 * pos] = sz;
        out[3 + 3 * pos] = first_base[uint64_t(blockIdx.y) * stride + i];
    }
}


// =====================================================================================================
// launchers
// =====================================================================================================
static int g_score_variant = 0;
static int g_part32 = 0;
static int g_score_oob = 0;  // k_score_part without a window test (ioc_ctx_create's probe passed, or IOC_SCORE_OOB=1)
// sharded merge: this rank scores the queries j with j % stride == offset 
// (per calling thread: two contexts driven from two threads do not see each other's setting; ioc_score resets both through a
// scope guard on every way out)
static thread_local int g_own_stride = 1, g_own_offset = 0;
static thread_local const uint32_t* g_keep_q = nullptr;  // per-query compaction threshold (fast mode; null: the uniform `keep`)

extern "C" {

void iock_set_score_variant(int v) { g_score_variant = v; }
void iock_set_part32(int v) { g_part32 = v; }
void iock_set_score_oob(int v) { g_score_oob = v; }
void iock_set_score_keep(const uint32_t* keep_q) { g_keep_q = keep_q; }
void iock_set_score_shard(int stride, int offset)
{
    g_own_stride = stride > 1 ? stride : 1;
    g_own_offset = stride > 1 ? offset : 0;
}

hipError_t iock_lds_oob_probe(hipStream_t st, uint32_t* d_result /* 2 words, zeroed here */, uint32_t* h_result)
{
    const unsigned grid = 4096;
    CK(hipMemsetAsync(d_result, 0, 8, st));
    const uint32_t sizes[2] = {12000u, 60000u};  // config 2's histogram (3000 targets) and a large batch's
    CK(hipFuncSetAttribute((const void*)k_lds_oob_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (uint32_t dyn : sizes) hipLaunchKernelGGL(k_lds_oob_probe, dim3(grid), dim3(IOC_BLOCK), dyn, st, dyn, d_result);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(h_result, d_result, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (h_result[1] != 2u * grid) h_result[0] |= 4u;  // the probe itself did not run to the end
    return hipSuccess;
}


hipError_t iock_score(hipStream_t st, int n, uint32_t L, const int64_t* off_fwd, const int64_t* off_rev,
                      const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift, const void* post_,
                      uint32_t range, uint32_t keep, uint32_t* cand_key, uint32_t* cand_size, uint32_t* cand_count,
                      unsigned long long* traversed, const uint8_t* audit_valid, unsigned long long* audit_sum,
                      uint32_t* part, uint32_t* top_all, int post16, uint32_t* pmins, uint32_t* pbnd)
{
    const uint32_t* post = (const uint32_t*)post_;
    const uint16_t* post_h = (const uint16_t*)post_;
    if (n <= 0) return hipSuccess;
    // sharded merge: one workgroup (or IOC_PARTS of them) per OWNED query; the others have no candidates here
    const int own_s = audit_valid ? 1 : g_own_stride, own_o = audit_valid ? 0 : g_own_offset;  // (an audit launch visits every query)
    const int nown = owned_count(0, n, own_s, own_o);
    if (nown != n) CK(hipMemsetAsync(cand_count, 0, size_t(n) * 4, st));
    if (nown <= 0) return hipSuccess;
    uint32_t tmax = L + uint32_t(n - 1);
    uint32_t r = tmax < range ? (tmax ? tmax : 1) : range;
    size_t lds = size_t(2) * r * 4;
    if (part && pmins && pbnd && tmax <= range && cap >= 1024) {
        const Epochs E = iock_epoch_bounds(L, uint32_t(n));
        if (lds > 40 * 1024) {
            CK(hipFuncSetAttribute((const void*)k_score_part<uint32_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            CK(hipFuncSetAttribute((const void*)k_score_part<uint16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            CK(hipFuncSetAttribute((const void*)k_score_part<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            CK(hipFuncSetAttribute((const void*)k_score_compact, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        }
        uint32_t* max_len = pbnd + size_t(n) * 2 * (IOC_PARTS + 1);  // one extra word behind the boundaries
        CK(hipMemsetAsync(max_len, g_part32 ? 0xFF : 0, 4, st));  // IOC_PART32=1 forces u32 partials (tests)
        hipLaunchKernelGGL(k_partition_mins, dim3(nown), dim3(IOC_BLOCK), 0, st, n, off_fwd, off_rev, mins, shift, pmins, pbnd,
                           max_len, own_s, own_o);
#define LAUNCH_PART(PT, OOB, PP)                                                                                          \
    hipLaunchKernelGGL((k_score_part<PT, OOB>), dim3(unsigned(nown) * IOC_PARTS), dim3(IOC_BLOCK), lds / 2, st, n, L, off_fwd, \
                       off_rev, pmins, pbnd, (const uint4*)rows, cap, shift, PP, part, E, traversed, max_len, uint32_t(lds / 2), own_s, own_o)
        if (post16 && g_score_oob && IOC_SCORE_OOB)
            LAUNCH_PART(uint16_t, true, post_h);
        else if (post16)
            LAUNCH_PART(uint16_t, false, post_h);
        else
            LAUNCH_PART(uint32_t, false, post);   // (u32 postings keep their window test: flat_traverse)
#undef LAUNCH_PART
        hipLaunchKernelGGL(k_score_compact, dim3(nown), dim3(IOC_BLOCK), lds, st, n, L, part, keep, cand_key, cand_size,
                           cand_count, audit_valid, audit_sum, top_all, max_len, own_s, own_o, audit_valid ? nullptr : g_keep_q);
        return hipGetLastError();
    }
    const Epochs E = iock_epoch_bounds(L, uint32_t(n));
#define LAUNCH_SCORE(V, PT, PP)                                                                                      \
    do {                                                                                                             \
        if (lds > 48 * 1024)                                                                                         \
            CK(hipFuncSetAttribute((const void*)k_score_t<V, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds))); \
        hipLaunchKernelGGL((k_score_t<V, PT>), dim3(nown), dim3(IOC_BLOCK), lds, st, n, L, off_fwd, off_rev, mins,   \
                           (const uint4*)rows, cap, shift, PP, range, keep, cand_key, cand_size, cand_count,         \
                           traversed, E, audit_valid, audit_sum, own_s, own_o, audit_valid ? nullptr : g_keep_q); \
    } while (0)
    if (post16) {
        LAUNCH_SCORE(0, uint16_t, post_h);
        return hipGetLastError();
    }
    switch (g_score_variant) {  // ablation builds for profiling only (IOC_SCORE_VARIANT); 0 = production
        case 1: LAUNCH_SCORE(1, uint32_t, post); break;
        case 2: LAUNCH_SCORE(2, uint32_t, post); break;
        case 3: LAUNCH_SCORE(3, uint32_t, post); break;
        case 4: LAUNCH_SCORE(4, uint32_t, post); break;
        case 5: LAUNCH_SCORE(5, uint32_t, post); break;
        case 6: LAUNCH_SCORE(6, uint32_t, post); break;
        case 7: LAUNCH_SCORE(7, uint32_t, post); break;
        default: LAUNCH_SCORE(0, uint32_t, post); break;
    }
    return hipGetLastError();
}


// non-empty entries of a query's hit table, as (index, Size, first Index) triples in any order: out[0] = count
__global__ void __launch_bounds__(256) k_query_compact(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ first,
                                                        uint32_t n2, uint32_t cap, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const uint32_t sz = hist[i];
    if (!sz) return;
    const uint32_t pos = atomicAdd(&out[0], 1u);
    if (pos < cap) {
        out[1 + 3 * pos] = i;
        out[2 + 3 * pos] = sz;
        out[3 + 3 * pos] = first[i];
    }
}

hipError_t iock_query_compact(hipStream_t st, const uint32_t* hist, const uint32_t* first, uint32_t n2, uint32_t cap, uint32_t* out)
{
    hipLaunchKernelGGL(k_query_compact, dim3((n2 + 255) / 256), dim3(256), 0, st, hist, first, n2, cap, out);
    return hipGetLastError();
}

hipError_t iock_query_table_many(hipStream_t st, int nq, const int32_t* qlist, uint64_t stride, uint32_t L, const int64_t* off_fwd,
                                 const int64_t* off_rev, const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift,
                                 const void* post, const uint8_t* valid, uint32_t* hist, uint32_t* first, int post16, uint32_t ccap,
                                 uint32_t* out)
{
    if (post16)
        hipLaunchKernelGGL(k_query_table_many<uint16_t>, dim3(16, unsigned(nq)), dim3(IOC_BLOCK), 0, st, qlist, stride, L, off_fwd, off_rev,
                           mins, (const uint4*)rows, cap, shift, (const uint16_t*)post, valid, hist, first);
    else
        hipLaunchKernelGGL(k_query_table_many<uint32_t>, dim3(16, unsigned(nq)), dim3(IOC_BLOCK), 0, st, qlist, stride, L, off_fwd, off_rev,
                           mins, (const uint4*)rows, cap, shift, (const uint32_t*)post, valid, hist, first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_query_compact_many, dim3(unsigned((stride + 255) / 256), unsigned(nq)), dim3(256), 0, st, qlist, stride, L, hist, first,
                       ccap, out);
    return hipGetLastError();
}

hipError_t iock_query_table(hipStream_t st, int j, uint32_t L, const int64_t* off_fwd, const int64_t* off_rev,
                            const uint32_t* mins, const void* rows, uint32_t cap, uint32_t shift, const void* post,
                            const uint8_t* valid, uint32_t* hist, uint32_t* first, int post16)
{
    if (post16)
        hipLaunchKernelGGL(k_query_table<uint16_t>, dim3(64), dim3(IOC_BLOCK), 0, st, j, L, off_fwd, off_rev, mins,
                           (const uint4*)rows, cap, shift, (const uint16_t*)post, valid, hist, first);
    else
        hipLaunchKernelGGL(k_query_table<uint32_t>, dim3(64), dim3(IOC_BLOCK), 0, st, j, L, off_fwd, off_rev, mins,
                           (const uint4*)rows, cap, shift, (const uint32_t*)post, valid, hist, first);
    return hipGetLastError();
}


}  // extern "C"

// (ioc_ctx_prewarm: makes the runtime load this file's code object now instead of at its first launch)
extern "C" hipError_t iock_warm_score()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_partition_mins));
}
