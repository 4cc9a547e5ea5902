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
//
// This file: the INDEX BUILD (distinct values, hash insert, scans, posting fill and sort, row packing, MinDB export).
// Scoring and the hit tables: ioc_score.hip.  Resolve (gap bounds, decide / evaluate / pick): ioc_resolve.hip.  Shared device
// helpers: ioc_kdev.h.
#include "ioc_kdev.h"

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
    if (m > pmax) return;  // (beyond this kernel's LDS: the host sends such a query through iock_distinct_long)
    uint32_t P = IOC_BLOCK;
    while (P < m) P <<= 1;
    if (P > pmax) P = pmax;  // (pmax <= IOC_BLOCK * IOC_DR_PER)
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
// launchers
// =====================================================================================================
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

hipError_t iock_distinct(hipStream_t st, int n, const int64_t* off_fwd, const uint32_t* mins, const int64_t* doff,
                         uint32_t* dvals, uint32_t* dcount, uint32_t pmax, int value_bits, uint32_t* pk, void* pv, int pv16, uint32_t target0,
                         uint32_t sentinel, int* pairs_written)
{
    if (pairs_written) *pairs_written = 0;
    if (n <= 0) return hipSuccess;
    size_t lds = size_t(pmax) * 4;
    const bool radix = !(getenv("IOC_DISTINCT_BITONIC") && atoi(getenv("IOC_DISTINCT_BITONIC")) == 1);  // (=1: round 1's bitonic network, for comparison)
    if (radix && IOC_BLOCK == 256) {
        // (a batch with a query beyond IOC_BLOCK * IOC_DR_PER values: this kernel takes the others, iock_distinct_long that one)
        const uint32_t pm = pmax < IOC_BLOCK ? IOC_BLOCK : std::min<uint32_t>(pmax, IOC_BLOCK * IOC_DR_PER);
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


}  // extern "C"

// (ioc_ctx_prewarm: makes the runtime load this file's code object now instead of at its first launch)
extern "C" hipError_t iock_warm_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_distinct_radix));
}
