// ioc_kernels.hip — CDNA4 (gfx950) kernels of the read->cluster assignment path.
//
// No MFMA anywhere: the path is integer hashing / gather / histogram work bounded by HBM + LDS
// (SURVEY.md §8(d)).  Wavefront = 64 lanes throughout; every kernel uses 256-thread workgroups
// (4 waves) so that LDS histograms of several queries are co-resident on one CU.
//
// Kernel            replaces (reference file:line)
// k_distinct        the dedupe of AddMinimizers (`cls > back()`), src/minimizer.cpp:31-42
// k_hash_insert*    MinimizerDB insert (unordered_map<unsigned, RepSet>), src/minimizer.h:60-61
// k_scan_*          (posting-list offsets: CSR instead of vector<unsigned> per key)
// k_fill_*          RepSet::emplace_back, src/minimizer.cpp:38-40
// k_score           GetMinimizerHits + ConsolidateMinimizerHits + Size of SortMinimizerHits,
//                   src/minimizer.cpp:44-76, src/cluster.cpp:609-636
// k_decide_scan / k_eval / k_decide_pick
//                   getBestClusterMapping + getMappedRatio, src/cluster.cpp:324-406
// k_query_table     the full hit map of one query (tie replay), src/minimizer.cpp:44-76
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "ioc_kernels.h"

#define IOC_BLOCK 256
#define IOC_WAVES (IOC_BLOCK / 64)
#define IOC_EMPTY 0xFFFFFFFFu
#ifndef IOC_FLAT_UNROLL
#define IOC_FLAT_UNROLL 4   // (8 until round 4: the long-chunk path inlined into k_score_part then costs 18 more spilled registers under its 64-register budget)
#endif
#ifndef IOC_FLAT_TAIL
#ifndef IOC_FLAT_TAIL
#define IOC_FLAT_TAIL 1
#endif
// IOC_FLAT_TAIL: steps per group in the tail of a chunk (flat_traverse_u16)
#endif
#define IOC_SHORT_LIST 192
#ifndef IOC_SCORE_OOB
#define IOC_SCORE_OOB 1  // k_score_part: the window test of a posting is the LDS allocation's own bounds check (see count_word_u16)
#endif
#ifndef IOC_SCORE_TRAV_CAPACITY
#define IOC_SCORE_TRAV_CAPACITY 0  // 1 (instrumentation builds): IOC_COUNT_TRAVERSED counts the posting SLOTS of the wave steps, filled or not
#endif
#define IOC_OOB_FAR_BASE 0x00100000u  // counter base of lanes past the end of a chunk in the OOB variant (1 MB: outside any LDS)
#ifndef IOC_SCORE_ABL
#define IOC_SCORE_ABL 0
#endif
#ifndef IOC_SCORE_OLD_TRAVERSE
#define IOC_SCORE_OLD_TRAVERSE 0  // 1: round 1's per-posting code in k_score_part (ablation builds)
#endif

namespace {

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & 63; }
__device__ __forceinline__ int wave_id() { return int(threadIdx.x) >> 6; }

__device__ __forceinline__ uint32_t hash_slot(uint32_t v, uint32_t shift)
{
    return (v * 0x9E3779B1u) >> shift;
}

// Rows of the index: {key, list offset, w2, w3}.  A list of >= IOC_EPOCH_LONG entries has w3 = 0x80000000 and w2 = its
// length.  A shorter one packs, next to its length (10 bits), where it can be CUT for a query that sees only the targets
// below T: f_i = ceil(#entries below the epoch boundary e_i / 8), 7 bits each, for the 7 boundaries e_1 < ... < e_7 that cut
// the target ids into 8 equal ranges — w2 = len | f1 << 10 | f2 << 17 | f3 << 24, w3 = f4 | f5 << 7 | f6 << 14 | f7 << 21.
// (Round 1 had 3 boundaries: a query then walked, on average, an eighth of every list beyond its window; now a sixteenth.)
// (IOC_EPOCHS, IOC_EPOCH_LONG, struct Epochs: ioc_kernels.h — the sorted index build fills the same fields)
// which field holds the cut of a query with window T: (word 0 = w2 / 1 = w3, shift); word 2 = no cut (T beyond e_7)
__device__ __forceinline__ void epoch_field(const Epochs& E, uint32_t T, uint32_t& word, uint32_t& shift)
{
    int f = IOC_EPOCHS;
#pragma unroll
    for (int i = IOC_EPOCHS - 1; i >= 0; --i)
        if (T <= E.e[i]) f = i;
    word = f < 3 ? 0u : f < IOC_EPOCHS ? 1u : 2u;
    shift = f < 3 ? 10u + 7u * uint32_t(f) : 7u * uint32_t(f - 3);
}
// visible length of a short list (info = {w2, w3}, len already decoded) under (word, shift) of epoch_field
__device__ __forceinline__ uint32_t epoch_cut(uint2 info, uint32_t len, uint32_t word, uint32_t shift)
{
    if (word == 2u) return len;
    const uint32_t f = ((word ? info.y : info.x) >> shift) & 127u;
    return min(len, f * 8u);
}

// Lookup in the packed rows — one 16-byte load per probe step.  cnt = the list's length, info = {w2, w3}.
__device__ __forceinline__ bool index_lookup(const uint4* __restrict__ rows, uint32_t cap, uint32_t shift,
                                             uint32_t v, uint32_t& off, uint32_t& cnt, uint2& info)
{
    if (v == IOC_EMPTY) {
        uint4 r = rows[cap];
        off = r.y;
        cnt = (r.w & 0x80000000u) ? r.z : (r.z & 1023u);
        info = make_uint2(r.z, r.w);
        return cnt != 0;
    }
    uint32_t h = hash_slot(v, shift);
    for (uint32_t step = 0; step < cap; ++step) {
        uint4 r = rows[h];
        if (r.x == v) {
            off = r.y;
            cnt = (r.w & 0x80000000u) ? r.z : (r.z & 1023u);
            info = make_uint2(r.z, r.w);
            return true;
        }
        if (r.x == IOC_EMPTY) return false;
        h = (h + 1) & (cap - 1);
    }
    return false;
}

// inclusive prefix sum over the 64 lanes: 4 DPP row shifts inside the rows of 16 lanes, then the two row broadcasts
// (lane 15 of a row to the next row, lane 31 to the upper half) — 6 data-parallel adds, no LDS crossbar (the
// __shfl_up form cost 5 VALU + 1 ds_bpermute per step)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#define IOC_DPP_ADD(ctrl, rmask) v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), ctrl, rmask, 0xF, false))
    IOC_DPP_ADD(0x111, 0xF);  // row_shr:1
    IOC_DPP_ADD(0x112, 0xF);  // row_shr:2
    IOC_DPP_ADD(0x114, 0xF);  // row_shr:4
    IOC_DPP_ADD(0x118, 0xF);  // row_shr:8
    IOC_DPP_ADD(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
    IOC_DPP_ADD(0x143, 0xC);  // row_bcast:31 into rows 2 and 3
#undef IOC_DPP_ADD
    return v;
}

// exclusive scan over the block; sh must hold IOC_WAVES words; two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* sh)
{
    uint32_t incl = wave_incl_scan(v);
    if (lane_id() == 63) sh[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < IOC_WAVES; ++w) {
        uint32_t s = sh[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

}  // namespace

// =====================================================================================================
// k_distinct: one workgroup per query; forward minimizer values -> LDS bitonic sort -> unique.
// dvals segment of query j starts at doff[j] (capacity nFwd[j]); dcount[j] = #distinct.
// =====================================================================================================
__global__ void __launch_bounds__(IOC_BLOCK)
k_distinct(int n, const int64_t* __restrict__ off_fwd, const uint32_t* __restrict__ mins,
           const int64_t* __restrict__ doff, uint32_t* __restrict__ dvals, uint32_t* __restrict__ dcount,
           uint32_t pmax)
{
    extern __shared__ uint32_t s[];  // pmax words
    __shared__ uint32_t sh[IOC_WAVES];
    int j = blockIdx.x;
    if (j >= n) return;
    int64_t b = off_fwd[j];
    uint32_t m = uint32_t(off_fwd[j + 1] - b);
    if (m == 0) {
        if (threadIdx.x == 0) dcount[j] = 0;
        return;
    }
    uint32_t P = 1;
    while (P < m) P <<= 1;
    if (P > pmax) P = pmax;  // host guarantees m <= pmax
    for (uint32_t i = threadIdx.x; i < P; i += IOC_BLOCK) s[i] = (i < m) ? mins[b + i] : IOC_EMPTY;
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += IOC_BLOCK) {
                uint32_t x = i ^ j2;
                if (x > i) {
                    uint32_t a = s[i], c = s[x];
                    bool asc = (i & k2) == 0;
                    if ((a > c) == asc) {
                        s[i] = c;
                        s[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // unique over the first m sorted entries (padding sorts to the end; equal-to-padding real
    // values are indistinguishable from it but then identical, so the prefix is the multiset).
    uint32_t base = 0;
    uint32_t* out = dvals + doff[j];
    for (uint32_t c = 0; c < m; c += IOC_BLOCK) {
        uint32_t i = c + threadIdx.x;
        uint32_t flag = (i < m) && (i == 0 || s[i] != s[i - 1]);
        uint32_t tot;
        uint32_t ex = block_excl_scan(flag, tot, sh);
        if (flag) out[base + ex] = s[i];
        base += tot;
    }
    if (threadIdx.x == 0) dcount[j] = base;
}

// =====================================================================================================
// k_distinct_radix: the same result as k_distinct for queries of up to 8192 forward minimizers, by an LSD radix sort
// (8-bit digits, ceil(2k / 8) passes) instead of a bitonic network: a thread keeps its 32 keys in registers, a pass is
// one LDS histogram per wave (plain ds_add), a scan over (digit, wave), and a stable scatter into LDS — the rank of a
// key among the equal digits of its 64-key chunk by an 8-ballot match-any, the chunk's base from the wave's running
// counter.  ~6 k instructions per wave where the network took ~35 k (91 compare-exchange stages over 8192 keys).
// =====================================================================================================
#define IOC_DR_PER 32
__global__ void __launch_bounds__(IOC_BLOCK)
k_distinct_radix(int n, const int64_t* __restrict__ off_fwd, const uint32_t* __restrict__ mins,
                 const int64_t* __restrict__ doff, uint32_t* __restrict__ dvals, uint32_t* __restrict__ dcount,
                 uint32_t pmax, int passes, uint32_t* __restrict__ pk, void* __restrict__ pv, int pv16, uint32_t target0, uint32_t sentinel)
{
    extern __shared__ uint32_t s[];  // pmax words
    __shared__ uint32_t hist[IOC_WAVES][256];
    __shared__ uint32_t sh[IOC_WAVES];
    const int j = blockIdx.x;
    if (j >= n) return;
    const int64_t b = off_fwd[j];
    const uint32_t m = uint32_t(off_fwd[j + 1] - b);
    if (m == 0) {
        if (threadIdx.x == 0) dcount[j] = 0;
        return;
    }
    uint32_t P = IOC_BLOCK;
    while (P < m) P <<= 1;
    if (P > pmax) P = pmax;  // host guarantees m <= pmax <= IOC_BLOCK * IOC_DR_PER
    const uint32_t per = P / IOC_BLOCK;  // keys per thread = 64-key chunks per wave
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6));
    const uint32_t wbase = wave * (P / IOC_WAVES) + uint32_t(lane);  // position of (wave, chunk c, lane) = wbase + 64 c
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t key[IOC_DR_PER];
#pragma unroll
    for (int c = 0; c < IOC_DR_PER; ++c) {
        key[c] = IOC_EMPTY;
        if (uint32_t(c) < per) {
            const uint32_t pos = wbase + 64u * uint32_t(c);
            if (pos < m) key[c] = mins[b + pos];
        }
    }
    for (int pass = 0; pass < passes; ++pass) {
        const uint32_t shift = 8u * uint32_t(pass);
        for (uint32_t i = threadIdx.x; i < IOC_WAVES * 256u; i += IOC_BLOCK) (&hist[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < IOC_DR_PER; ++c)
            if (uint32_t(c) < per) atomicAdd(&hist[wave][(key[c] >> shift) & 255u], 1u);
        __syncthreads();
        {   // hist[w][d] := number of keys with a smaller digit, or the same digit in an earlier wave
            const uint32_t d = threadIdx.x;  // IOC_BLOCK == 256 digits
            uint32_t h[IOC_WAVES], tot = 0;
#pragma unroll
            for (int w = 0; w < IOC_WAVES; ++w) {
                h[w] = hist[w][d];
                tot += h[w];
            }
            uint32_t all;
            uint32_t ex = block_excl_scan(tot, all, sh);
#pragma unroll
            for (int w = 0; w < IOC_WAVES; ++w) {
                hist[w][d] = ex;
                ex += h[w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < IOC_DR_PER; ++c) {
            if (uint32_t(c) < per) {
                const uint32_t d = (key[c] >> shift) & 255u;
                unsigned long long peers = ~0ull;
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const bool one = (d >> bit) & 1u;
                    const unsigned long long bm = __ballot(one);
                    peers &= one ? bm : ~bm;
                }
                const uint32_t base = hist[wave][d];
                const uint32_t rank = uint32_t(__popcll(peers & lt_mask));
                __builtin_amdgcn_wave_barrier();  // every lane has read the counter before its digit's first lane moves it on
                if (rank == 0) hist[wave][d] = base + uint32_t(__popcll(peers));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                s[base + rank] = key[c];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < IOC_DR_PER; ++c)
            if (uint32_t(c) < per) key[c] = s[wbase + 64u * uint32_t(c)];
        __syncthreads();
    }
    // (s holds the keys in ascending order — of their low 8 * passes bits, which are all the bits a value has; the padding
    // 0xFFFFFFFF started behind every value and a stable sort leaves it behind the values it ties with)
    uint32_t base = 0;
    uint32_t* out = dvals + doff[j];
    // (pk: the sorted index build's (value, target) pairs of this query, written here instead of by a kernel of their own; the
    // unused tail of the query's stretch carries the sentinel key)
    uint32_t* pko = pk ? pk + doff[j] : nullptr;
    for (uint32_t c = 0; c < m; c += IOC_BLOCK) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t flag = (i < m) && (i == 0 || s[i] != s[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan(flag, tot, sh);
        if (flag) {
            out[base + ex] = s[i];
            if (pko) pko[base + ex] = s[i];
        }
        base += tot;
    }
    if (threadIdx.x == 0) dcount[j] = base;
    if (pko) {
        for (uint32_t d = base + threadIdx.x; d < m; d += IOC_BLOCK) pko[d] = sentinel;
        const uint32_t t = target0 + uint32_t(j);
        if (pv16) {
            uint16_t* o = static_cast<uint16_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += IOC_BLOCK) o[d] = uint16_t(t);
        } else {
            uint32_t* o = static_cast<uint32_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += IOC_BLOCK) o[d] = t;
        }
    }
}

// =====================================================================================================
// hash build
// =====================================================================================================
__device__ __forceinline__ uint32_t hash_insert(uint32_t* __restrict__ keys, uint32_t cap, uint32_t shift,
                                                uint32_t v)
{
    if (v == IOC_EMPTY) return cap;
    uint32_t h = hash_slot(v, shift);
    for (uint32_t step = 0; step < cap; ++step) {
        uint32_t k = keys[h];
        if (k == v) return h;
        if (k == IOC_EMPTY) {
            uint32_t old = atomicCAS(&keys[h], IOC_EMPTY, v);
            if (old == IOC_EMPTY || old == v) return h;
        }
        h = (h + 1) & (cap - 1);
    }
    return cap + 1;  // table full (host sizes the table so that this cannot happen)
}

#define IOC_INS_ILP 4
__global__ void __launch_bounds__(IOC_BLOCK)
k_hash_insert_queries(int n, const int64_t* __restrict__ doff, const uint32_t* __restrict__ dvals,
                      const uint32_t* __restrict__ dcount, uint32_t* __restrict__ keys, uint32_t cap,
                      uint32_t shift, uint32_t* __restrict__ cnt, uint32_t* __restrict__ dslot,
                      uint32_t* __restrict__ dpos, uint32_t* __restrict__ err)
{
    int j = blockIdx.x;
    if (j >= n) return;
    int64_t b = doff[j];
    uint32_t m = dcount[j];
    // IOC_INS_ILP values per thread in flight: the first probe of each (almost always a hit once a few
    // queries have been inserted) and the returning atomicAdd are issued back to back
    for (uint32_t d0 = threadIdx.x; d0 < m; d0 += IOC_BLOCK * IOC_INS_ILP) {
        uint32_t v[IOC_INS_ILP], h[IOC_INS_ILP], k0[IOC_INS_ILP], slot[IOC_INS_ILP], pos[IOC_INS_ILP];
        bool in[IOC_INS_ILP];
#pragma unroll
        for (int u = 0; u < IOC_INS_ILP; ++u) {
            const uint32_t d = d0 + uint32_t(u) * IOC_BLOCK;
            in[u] = d < m;
            v[u] = in[u] ? dvals[b + d] : 0u;
            h[u] = hash_slot(v[u], shift);
        }
#pragma unroll
        for (int u = 0; u < IOC_INS_ILP; ++u) k0[u] = (in[u] && v[u] != IOC_EMPTY) ? keys[h[u]] : 0u;
#pragma unroll
        for (int u = 0; u < IOC_INS_ILP; ++u) {
            slot[u] = cap + 1;
            if (in[u]) slot[u] = (v[u] != IOC_EMPTY && k0[u] == v[u]) ? h[u] : hash_insert(keys, cap, shift, v[u]);
        }
#pragma unroll
        for (int u = 0; u < IOC_INS_ILP; ++u) {
            pos[u] = 0;
            if (in[u]) {
                if (slot[u] > cap) {
                    atomicAdd(err, 1u);
                    slot[u] = cap;
                } else {
                    pos[u] = atomicAdd(&cnt[slot[u]], 1u);  // position inside the posting list (after the left part)
                }
            }
        }
#pragma unroll
        for (int u = 0; u < IOC_INS_ILP; ++u) {
            const uint32_t d = d0 + uint32_t(u) * IOC_BLOCK;
            if (in[u]) {
                dslot[b + d] = slot[u];
                dpos[b + d] = pos[u];
            }
        }
    }
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_hash_insert_left(int64_t nkeys, const uint32_t* __restrict__ lkeys, const int64_t* __restrict__ loffs,
                   uint32_t* __restrict__ keys, uint32_t cap, uint32_t shift, uint32_t* __restrict__ cnt,
                   uint32_t* __restrict__ lslot, uint32_t* __restrict__ err)
{
    int64_t i = int64_t(blockIdx.x) * IOC_BLOCK + threadIdx.x;
    if (i >= nkeys) return;
    uint32_t slot = hash_insert(keys, cap, shift, lkeys[i]);
    if (slot > cap) {
        atomicAdd(err, 1u);
        slot = cap;
    } else {
        atomicAdd(&cnt[slot], uint32_t(loffs[i + 1] - loffs[i]));
    }
    lslot[i] = slot;
}

// ---- 3-phase exclusive scan over u32 (n up to 2^31): block = 1024 elements -------------------------
#define IOC_SCAN_ELEMS 1024
__global__ void __launch_bounds__(IOC_BLOCK)
k_scan_reduce(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ block_sums, uint32_t rmask)
{
    __shared__ uint32_t sh[IOC_WAVES];
    int64_t base = int64_t(blockIdx.x) * IOC_SCAN_ELEMS;
    uint32_t v = 0;
#pragma unroll
    for (int r = 0; r < IOC_SCAN_ELEMS / IOC_BLOCK; ++r) {
        int64_t i = base + r * IOC_BLOCK + threadIdx.x;
        if (i < n) v += (in[i] + rmask) & ~rmask;
    }
    uint32_t tot;
    block_excl_scan(v, tot, sh);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place, total -> block_sums[nb]
__global__ void __launch_bounds__(IOC_BLOCK)
k_scan_sums(uint32_t* __restrict__ block_sums, int64_t nb)
{
    __shared__ uint32_t sh[IOC_WAVES];
    uint32_t carry = 0;
    for (int64_t c = 0; c < nb; c += IOC_BLOCK) {
        int64_t i = c + threadIdx.x;
        uint32_t v = (i < nb) ? block_sums[i] : 0;
        uint32_t tot;
        uint32_t ex = block_excl_scan(v, tot, sh);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_scan_apply(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ block_sums,
             uint32_t* __restrict__ out, uint32_t rmask)
{
    __shared__ uint32_t sh[IOC_WAVES];
    int64_t base = int64_t(blockIdx.x) * IOC_SCAN_ELEMS;
    uint32_t carry = block_sums[blockIdx.x];
    // each thread owns 4 consecutive elements so that the scan order is the element order
    int64_t i0 = base + int64_t(threadIdx.x) * (IOC_SCAN_ELEMS / IOC_BLOCK);
    uint32_t v[IOC_SCAN_ELEMS / IOC_BLOCK];
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < IOC_SCAN_ELEMS / IOC_BLOCK; ++r) {
        v[r] = (i0 + r < n) ? ((in[i0 + r] + rmask) & ~rmask) : 0;
        s += v[r];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan(s, tot, sh) + carry;
#pragma unroll
    for (int r = 0; r < IOC_SCAN_ELEMS / IOC_BLOCK; ++r) {
        if (i0 + r < n) out[i0 + r] = ex;
        ex += v[r];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == IOC_BLOCK - 1) {
        // total goes to out[n]: last thread's running value covers every element < n
        out[n] = ex;
    }
}

// ---- fill ------------------------------------------------------------------------------------------
// Postings are stored as PT = uint16_t when every target id fits (L + N <= 65535: 8 postings per 16-byte
// unit, half the bytes), else uint32_t.
template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_fill_left(int64_t nkeys, const int64_t* __restrict__ loffs, const uint32_t* __restrict__ lpost,
            const uint32_t* __restrict__ lslot, const uint32_t* __restrict__ off, PT* __restrict__ post)
{
    // one wave per key
    int64_t key = (int64_t(blockIdx.x) * IOC_BLOCK + threadIdx.x) >> 6;
    if (key >= nkeys) return;
    uint32_t slot = lslot[key];
    int64_t b = loffs[key];
    uint32_t m = uint32_t(loffs[key + 1] - b);
    uint32_t o = off[slot];
    for (uint32_t t = lane_id(); t < m; t += 64) post[o + t] = PT(lpost[b + t]);
}

template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_fill_queries(int n, uint32_t L, const int64_t* __restrict__ doff, const uint32_t* __restrict__ dcount,
               const uint32_t* __restrict__ dslot, const uint32_t* __restrict__ dpos,
               const uint32_t* __restrict__ off, PT* __restrict__ post)
{
    int j = blockIdx.x;
    if (j >= n) return;
    int64_t b = doff[j];
    uint32_t m = dcount[j];
    for (uint32_t d = threadIdx.x; d < m; d += IOC_BLOCK) {
        post[off[dslot[b + d]] + dpos[b + d]] = PT(L + uint32_t(j));
    }
}

// Posting lists must be ascending in target id (the reference keeps RepSet ascending,
// src/minimizer.cpp:38-40); the atomic fill leaves the query part of each list unordered.
// One wave per slot: rank-by-counting for lists <= 64; longer lists go through a per-wave LDS bitmap
// (the query part holds distinct integers in [L, L+n)).
template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_sort_lists(uint32_t nslots, const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt,
             PT* __restrict__ post, uint32_t L, uint32_t words_per_wave, Epochs E, uint2* __restrict__ qinfo, int sorted)
{
    extern __shared__ uint32_t sbits[];  // IOC_WAVES * words_per_wave
    const uint32_t gw = (blockIdx.x * IOC_BLOCK + threadIdx.x) >> 6;  // global wave id
    const uint32_t nw = (gridDim.x * IOC_BLOCK) >> 6;
    const int lane = lane_id();
    uint32_t* bits = sbits + size_t(wave_id()) * words_per_wave;
    for (uint32_t slot = gw; slot < nslots; slot += nw) {
        const uint32_t c = cnt[slot];
        // epoch cuts (see index_lookup): per boundary the number of entries below it, in units of 8 postings
        // (the counts do not depend on the order of the list, which is only sorted below)
        {
            uint2 info = make_uint2(c, 0x80000000u);
            if (c < IOC_EPOCH_LONG) {
                const uint32_t o0 = off[slot];
                uint32_t b[IOC_EPOCHS];  // wave-uniform: one compare + ballot + scalar popcount per boundary and 64 entries
#pragma unroll
                for (int i = 0; i < IOC_EPOCHS; ++i) b[i] = 0;
                for (uint32_t t0 = 0; t0 < c; t0 += 64) {
                    const uint32_t t = t0 + uint32_t(lane);
                    const uint32_t v = t < c ? uint32_t(post[o0 + t]) : IOC_EMPTY;
#pragma unroll
                    for (int i = 0; i < IOC_EPOCHS; ++i) b[i] += uint32_t(__popcll(__ballot(v < E.e[i])));
                }
#pragma unroll
                for (int i = 0; i < IOC_EPOCHS; ++i) b[i] = (b[i] + 7u) >> 3;
                info.x = c | (b[0] << 10) | (b[1] << 17) | (b[2] << 24);
                info.y = b[3] | (b[4] << 7) | (b[5] << 14) | (b[6] << 21);
            }
            if (lane == 0) qinfo[slot] = info;
        }
        if (c < 2 || sorted) continue;  // (sorted: the lists of ioc_build_sort.hip come out ascending)
        const uint32_t o = off[slot];
        // the left part (values < L) was copied first and is already ascending: it is a prefix by position
        uint32_t a = 0, b2 = c;
        while (a < b2) {
            uint32_t mid = (a + b2) >> 1;
            if (post[o + mid] < L) a = mid + 1; else b2 = mid;
        }
        const uint32_t m = c - a;
        if (m < 2) continue;
        PT* p = post + o + a;
        if (m <= 64) {
            uint32_t x = (uint32_t(lane) < m) ? p[lane] : IOC_EMPTY;
            uint32_t rank = 0;
            for (uint32_t t = 0; t < m; ++t) {
                uint32_t y = __shfl(x, int(t));
                rank += (y < x);
            }
            if (uint32_t(lane) < m) p[rank] = PT(x);
        } else {
            for (uint32_t wd = lane; wd < words_per_wave; wd += 64) bits[wd] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (uint32_t t = lane; t < m; t += 64) {
                uint32_t id = p[t] - L;
                atomicOr(&bits[id >> 5], 1u << (id & 31));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            uint32_t base = 0;
            for (uint32_t w0 = 0; w0 < words_per_wave; w0 += 64) {
                uint32_t wd = w0 + lane;
                uint32_t bw = (wd < words_per_wave) ? bits[wd] : 0;
                uint32_t pc = __popc(bw);
                uint32_t incl = wave_incl_scan(pc);
                uint32_t ex = base + incl - pc;
                while (bw) {
                    uint32_t bit = __ffs(bw) - 1;
                    bw &= bw - 1;
                    p[ex++] = PT(L + wd * 32 + bit);
                }
                base += __shfl(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
}

__global__ void __launch_bounds__(IOC_BLOCK)
k_pack_rows(uint32_t nslots, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ off,
            const uint32_t* __restrict__ cnt, const uint2* __restrict__ qinfo, uint4* __restrict__ rows)
{
    uint32_t s = blockIdx.x * IOC_BLOCK + threadIdx.x;
    if (s >= nslots) return;
    const uint2 q = qinfo[s];  // {length | cuts, cuts} or {length, long-list flag}: see index_lookup
    rows[s] = make_uint4(keys[s], off[s], q.x, q.y);
}

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

// Sharded merge (ioc_set_shard): this rank's queries are j = own_offset (mod own_stride); they are DENSE in blockIdx (a
// strided blockIdx would put every owned workgroup on the same XCD: workgroups are dealt round-robin over the 8 XCDs).
// The b-th owned query counted from the top of [0, n) (scoring visits the long target ranges first) / from `from` upwards.
__device__ __forceinline__ int owned_from_top(int n, int b, int stride, int offset)
{
    if (stride <= 1) return n - 1 - b;
    const int top = (n - 1) - (((n - 1) - offset) % stride + stride) % stride;  // largest j <= n - 1 with j % stride == offset
    return top - b * stride;
}
__device__ __forceinline__ int owned_from(int from, int b, int stride, int offset)
{
    if (stride <= 1) return from + b;
    const int j0 = from + ((offset - from) % stride + stride) % stride;  // smallest j >= from with j % stride == offset
    return j0 + b * stride;
}
static inline int owned_count(int from, int n, int stride, int offset)
{
    if (stride <= 1) return n > from ? n - from : 0;
    const int j0 = from + ((offset - from) % stride + stride) % stride;
    return j0 < n ? (n - j0 + stride - 1) / stride : 0;
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
#define CK(x)                     \
    do {                          \
        hipError_t e_ = (x);      \
        if (e_ != hipSuccess) return e_; \
    } while (0)

static int g_score_variant = 0;
static int g_part32 = 0;
static int g_score_oob = 0;  // k_score_part without a window test (ioc_ctx_create's probe passed, or IOC_SCORE_OOB=1)
// sharded merge: this rank scores the queries j with j % stride == offset 
// (per calling thread: two contexts driven from two threads do not see each other's setting; ioc_score resets both through a
// scope guard on every way out)
static thread_local int g_own_stride = 1, g_own_offset = 0;
static thread_local const uint32_t* g_keep_q = nullptr;  // per-query compaction threshold (fast mode; null: the uniform `keep`)

namespace {
// ---- MinDB export (ioc_index_export): the posting lists restricted to the targets that ARE clusters, with final ids ----
// cid[t - L] = final cluster id of query t - L if it opened a cluster, -1 otherwise; left targets keep their ids.
// One wave per slot; the order inside a list (ascending targets) is kept, and final ids ascend with the targets.
template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_export_count(uint32_t nslots, const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt, const PT* __restrict__ post,
               uint32_t L, const int32_t* __restrict__ cid, uint32_t* __restrict__ out_cnt)
{
    const uint32_t gw = (blockIdx.x * IOC_BLOCK + threadIdx.x) >> 6, nw = (gridDim.x * IOC_BLOCK) >> 6;
    const int lane = lane_id();
    for (uint32_t slot = gw; slot < nslots; slot += nw) {
        const uint32_t c = cnt[slot], o = off[slot];
        uint32_t k = 0;
        for (uint32_t t0 = 0; t0 < c; t0 += 64) {
            const uint32_t t = t0 + uint32_t(lane);
            bool keep = false;
            if (t < c) {
                const uint32_t tg = post[o + t];
                keep = tg < L || cid[tg - L] >= 0;
            }
            k += uint32_t(__popcll(__ballot(keep)));
        }
        if (lane == 0) out_cnt[slot] = k;
    }
}

template <typename PT>
__global__ void __launch_bounds__(IOC_BLOCK)
k_export_fill(uint32_t nslots, const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt, const PT* __restrict__ post,
              uint32_t L, const int32_t* __restrict__ cid, const uint32_t* __restrict__ out_cnt, const int64_t* __restrict__ out_off,
              uint32_t* __restrict__ out)
{
    const uint32_t gw = (blockIdx.x * IOC_BLOCK + threadIdx.x) >> 6, nw = (gridDim.x * IOC_BLOCK) >> 6;
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t slot = gw; slot < nslots; slot += nw) {
        if (out_cnt[slot] == 0) continue;
        const uint32_t c = cnt[slot], o = off[slot];
        int64_t w = out_off[slot];
        for (uint32_t t0 = 0; t0 < c; t0 += 64) {
            const uint32_t t = t0 + uint32_t(lane);
            int32_t id = -1;
            if (t < c) {
                const uint32_t tg = post[o + t];
                id = tg < L ? int32_t(tg) : cid[tg - L];
            }
            const unsigned long long m = __ballot(id >= 0);
            if (id >= 0) out[w + __popcll(m & lt_mask)] = uint32_t(id);
            w += __popcll(m);
        }
    }
}

}  // namespace

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


hipError_t iock_distinct(hipStream_t st, int n, const int64_t* off_fwd, const uint32_t* mins, const int64_t* doff,
                         uint32_t* dvals, uint32_t* dcount, uint32_t pmax, int value_bits, uint32_t* pk, void* pv, int pv16, uint32_t target0,
                         uint32_t sentinel, int* pairs_written)
{
    if (pairs_written) *pairs_written = 0;
    if (n <= 0) return hipSuccess;
    size_t lds = size_t(pmax) * 4;
    const bool radix = !(getenv("IOC_DISTINCT_BITONIC") && atoi(getenv("IOC_DISTINCT_BITONIC")) == 1);  // (=1: round 1's bitonic network, for comparison)
    if (radix && pmax <= IOC_BLOCK * IOC_DR_PER && IOC_BLOCK == 256) {
        const uint32_t pm = pmax < IOC_BLOCK ? IOC_BLOCK : pmax;
        lds = size_t(pm) * 4;
        const int bits = value_bits < 1 ? 32 : (value_bits > 32 ? 32 : value_bits);
        if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k_distinct_radix, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(k_distinct_radix, dim3(n), dim3(IOC_BLOCK), lds, st, n, off_fwd, mins, doff, dvals, dcount, pm, (bits + 7) / 8, pk, pv, pv16,
                           target0, sentinel);
        if (pairs_written && pk) *pairs_written = 1;
        return hipGetLastError();
    }
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k_distinct, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_distinct, dim3(n), dim3(IOC_BLOCK), lds, st, n, off_fwd, mins, doff, dvals, dcount, pmax);
    return hipGetLastError();
}

hipError_t iock_hash_insert_queries(hipStream_t st, int n, const int64_t* doff, const uint32_t* dvals,
                                    const uint32_t* dcount, uint32_t* keys, uint32_t cap, uint32_t shift,
                                    uint32_t* cnt, uint32_t* dslot, uint32_t* dpos, uint32_t* err)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hash_insert_queries, dim3(n), dim3(IOC_BLOCK), 0, st, n, doff, dvals, dcount, keys, cap,
                       shift, cnt, dslot, dpos, err);
    return hipGetLastError();
}

hipError_t iock_hash_insert_left(hipStream_t st, int64_t nkeys, const uint32_t* lkeys, const int64_t* loffs,
                                 uint32_t* keys, uint32_t cap, uint32_t shift, uint32_t* cnt, uint32_t* lslot,
                                 uint32_t* err)
{
    if (nkeys <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hash_insert_left, dim3((unsigned)((nkeys + IOC_BLOCK - 1) / IOC_BLOCK)), dim3(IOC_BLOCK), 0,
                       st, nkeys, lkeys, loffs, keys, cap, shift, cnt, lslot, err);
    return hipGetLastError();
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total. scratch: ceil(n/1024)+1 words.
hipError_t iock_exclusive_scan(hipStream_t st, const uint32_t* in, int64_t n, uint32_t* out, uint32_t* scratch,
                               uint32_t round_mask)
{
    if (n <= 0) return hipSuccess;
    int64_t nb = (n + IOC_SCAN_ELEMS - 1) / IOC_SCAN_ELEMS;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(IOC_BLOCK), 0, st, in, n, scratch, round_mask);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(IOC_BLOCK), 0, st, scratch, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(IOC_BLOCK), 0, st, in, n, scratch, out, round_mask);
    return hipGetLastError();
}

hipError_t iock_fill_left(hipStream_t st, int64_t nkeys, const int64_t* loffs, const uint32_t* lpost,
                          const uint32_t* lslot, const uint32_t* off, void* post, int post16)
{
    if (nkeys <= 0) return hipSuccess;
    int64_t threads = nkeys * 64;
    const dim3 grid((unsigned)((threads + IOC_BLOCK - 1) / IOC_BLOCK));
    if (post16)
        hipLaunchKernelGGL(k_fill_left<uint16_t>, grid, dim3(IOC_BLOCK), 0, st, nkeys, loffs, lpost, lslot, off,
                           (uint16_t*)post);
    else
        hipLaunchKernelGGL(k_fill_left<uint32_t>, grid, dim3(IOC_BLOCK), 0, st, nkeys, loffs, lpost, lslot, off,
                           (uint32_t*)post);
    return hipGetLastError();
}

hipError_t iock_fill_queries(hipStream_t st, int n, uint32_t L, const int64_t* doff, const uint32_t* dcount,
                             const uint32_t* dslot, const uint32_t* dpos, const uint32_t* off, void* post, int post16)
{
    if (n <= 0) return hipSuccess;
    if (post16)
        hipLaunchKernelGGL(k_fill_queries<uint16_t>, dim3(n), dim3(IOC_BLOCK), 0, st, n, L, doff, dcount, dslot, dpos,
                           off, (uint16_t*)post);
    else
        hipLaunchKernelGGL(k_fill_queries<uint32_t>, dim3(n), dim3(IOC_BLOCK), 0, st, n, L, doff, dcount, dslot, dpos,
                           off, (uint32_t*)post);
    return hipGetLastError();
}

Epochs iock_epoch_bounds(uint32_t L, uint32_t n);
static Epochs epoch_bounds(uint32_t L, uint32_t n) { return iock_epoch_bounds(L, n); }
Epochs iock_epoch_bounds(uint32_t L, uint32_t n)
{
    Epochs E;
    for (int i = 0; i < IOC_EPOCHS; ++i) E.e[i] = L + uint32_t((uint64_t(n) * uint64_t(i + 1) + IOC_EPOCHS) / (IOC_EPOCHS + 1));
    return E;
}

hipError_t iock_sort_lists(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, void* post,
                           uint32_t L, uint32_t n, uint32_t nblocks, uint32_t* qinfo, int post16, int sorted)
{
    const Epochs E = epoch_bounds(L, n);
    uint32_t words = (n + 31) / 32;
    if (words == 0) words = 1;
    size_t lds = size_t(IOC_WAVES) * words * 4;
    if (post16) {
        if (lds > 48 * 1024)
            CK(hipFuncSetAttribute((const void*)k_sort_lists<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(k_sort_lists<uint16_t>, dim3(nblocks), dim3(IOC_BLOCK), lds, st, nslots, off, cnt,
                           (uint16_t*)post, L, words, E, (uint2*)qinfo, sorted);
    } else {
        if (lds > 48 * 1024)
            CK(hipFuncSetAttribute((const void*)k_sort_lists<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(k_sort_lists<uint32_t>, dim3(nblocks), dim3(IOC_BLOCK), lds, st, nslots, off, cnt,
                           (uint32_t*)post, L, words, E, (uint2*)qinfo, sorted);
    }
    return hipGetLastError();
}

// list e of the gather: words [src[e], src[e] + len[e]) of the source arrays to dst[e] of the destination arrays
__global__ void __launch_bounds__(IOC_BLOCK)
k_gather_lists(uint32_t nlists, const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const uint32_t* __restrict__ len,
               const uint32_t* __restrict__ smin, const uint32_t* __restrict__ spos, uint32_t* __restrict__ dmin,
               uint32_t* __restrict__ dpos)
{
    for (uint32_t e = blockIdx.x; e < nlists; e += gridDim.x) {
        const int64_t a = src[e], b = dst[e];
        const uint32_t n = len[e];
        for (uint32_t t = threadIdx.x; t < n; t += IOC_BLOCK) {
            dmin[b + t] = smin[a + t];
            dpos[b + t] = spos[a + t];
        }
    }
}

hipError_t iock_export_count(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, const void* post, int post16,
                             uint32_t L, const int32_t* cid, uint32_t* out_cnt)
{
    const unsigned nb = 2048;
    if (post16)
        hipLaunchKernelGGL(k_export_count<uint16_t>, dim3(nb), dim3(IOC_BLOCK), 0, st, nslots, off, cnt, (const uint16_t*)post, L, cid, out_cnt);
    else
        hipLaunchKernelGGL(k_export_count<uint32_t>, dim3(nb), dim3(IOC_BLOCK), 0, st, nslots, off, cnt, (const uint32_t*)post, L, cid, out_cnt);
    return hipGetLastError();
}

hipError_t iock_export_fill(hipStream_t st, uint32_t nslots, const uint32_t* off, const uint32_t* cnt, const void* post, int post16,
                            uint32_t L, const int32_t* cid, const uint32_t* out_cnt, const int64_t* out_off, uint32_t* out)
{
    const unsigned nb = 2048;
    if (post16)
        hipLaunchKernelGGL(k_export_fill<uint16_t>, dim3(nb), dim3(IOC_BLOCK), 0, st, nslots, off, cnt, (const uint16_t*)post, L, cid, out_cnt,
                           out_off, out);
    else
        hipLaunchKernelGGL(k_export_fill<uint32_t>, dim3(nb), dim3(IOC_BLOCK), 0, st, nslots, off, cnt, (const uint32_t*)post, L, cid, out_cnt,
                           out_off, out);
    return hipGetLastError();
}

hipError_t iock_gather_lists(hipStream_t st, uint32_t nlists, const int64_t* src, const int64_t* dst, const uint32_t* len,
                             const uint32_t* smin, const uint32_t* spos, uint32_t* dmin, uint32_t* dpos)
{
    if (nlists == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_lists, dim3(nlists < 65535u ? nlists : 65535u), dim3(IOC_BLOCK), 0, st, nlists, src, dst, len, smin,
                       spos, dmin, dpos);
    return hipGetLastError();
}

// several buffers filled by ONE launch (the index build clears five: each memset is a launch of its own otherwise)
struct FillSegs {
    uint32_t* p[IOC_FILL_SEGS];
    unsigned long long words[IOC_FILL_SEGS];  // 32-bit words
    uint32_t value[IOC_FILL_SEGS];
    unsigned long long first_block[IOC_FILL_SEGS + 1];  // blocks are dealt out in proportion to the segments' sizes
};
constexpr unsigned long long FILL_WORDS_PER_BLOCK = 256ull * 4ull * 8ull;  // 256 threads x uint4 x 8
__global__ void __launch_bounds__(256) k_fill_multi(FillSegs f)
{
    int sgm = 0;
#pragma unroll
    for (int x = 1; x < IOC_FILL_SEGS; ++x)
        if (blockIdx.x >= f.first_block[x]) sgm = x;
    const unsigned long long b = blockIdx.x - f.first_block[sgm];
    uint32_t* p = f.p[sgm];
    const unsigned long long nw = f.words[sgm];
    const uint32_t v = f.value[sgm];
    const unsigned long long w0 = b * FILL_WORDS_PER_BLOCK, w1 = min(nw, w0 + FILL_WORDS_PER_BLOCK);
    // (the buffers are 256-byte aligned device allocations: uint4 stores over whole quads, words behind them)
    const unsigned long long q0 = w0 / 4, q1 = w1 / 4;
    for (unsigned long long q = q0 + threadIdx.x; q < q1; q += 256) reinterpret_cast<uint4*>(p)[q] = uint4{v, v, v, v};
    for (unsigned long long w = q1 * 4 + threadIdx.x; w < w1; w += 256) p[w] = v;
}

hipError_t iock_fill_multi(hipStream_t st, int nseg, void* const* ptrs, const size_t* bytes, const uint32_t* values)
{
    if (nseg < 1 || nseg > IOC_FILL_SEGS) return hipErrorInvalidValue;
    FillSegs f{};
    unsigned long long blocks = 0;
    for (int x = 0; x < IOC_FILL_SEGS; ++x) {
        f.first_block[x] = blocks;
        if (x < nseg) {
            if ((bytes[x] & 3u) || (reinterpret_cast<uintptr_t>(ptrs[x]) & 15u)) return hipErrorInvalidValue;
            f.p[x] = static_cast<uint32_t*>(ptrs[x]);
            f.words[x] = bytes[x] / 4;
            f.value[x] = values[x];
            blocks += (f.words[x] + FILL_WORDS_PER_BLOCK - 1) / FILL_WORDS_PER_BLOCK;
        }
    }
    f.first_block[IOC_FILL_SEGS] = blocks;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fill_multi, dim3(unsigned(blocks)), dim3(256), 0, st, f);
    return hipGetLastError();
}

hipError_t iock_pack_rows(hipStream_t st, uint32_t nslots, const uint32_t* keys, const uint32_t* off,
                          const uint32_t* cnt, const uint32_t* qinfo, void* rows)
{
    hipLaunchKernelGGL(k_pack_rows, dim3((nslots + IOC_BLOCK - 1) / IOC_BLOCK), dim3(IOC_BLOCK), 0, st, nslots, keys,
                       off, cnt, (const uint2*)qinfo, (uint4*)rows);
    return hipGetLastError();
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
        const Epochs E = epoch_bounds(L, uint32_t(n));
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
    const Epochs E = epoch_bounds(L, uint32_t(n));
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

size_t iock_decide_args_size() { return sizeof(DecideArgs); }

}  // extern "C"
