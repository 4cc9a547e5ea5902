// ioc_build_sort.hip — the index build WITHOUT global atomics.
//
// AddMinimizers (src/minimizer.cpp:31-42) turns the queries' distinct forward minimizers into posting lists: a transpose of
// the sparse (query, value) matrix.  The first build of this path did it with one device-scope atomic per (value, query) — a
// counter per hash slot that also hands out the position in the list — and the chip does 27 G random atomics a second whatever
// they return (tools/micro/atomics.hip): 0.46 ms for the 12.3 M pairs of the 3000-read / 50 Mb batch before anything else.
// A stable LSD radix sort of the pairs by value does the same transpose with plain loads and stores (rocPRIM: 0.28 ms for the
// same pairs, tools/micro/sortpairs.hip), and because the pairs go in ordered by target — the left MinDB's postings first,
// then query by query — every list comes out ascending: the per-list sort of the atomic build is not needed either.
//
//   pairs     [left postings (key of their list, cluster id)] [query j's distinct values, capacity of its forward list: the
//             unused tail holds a sentinel key above every value]
//   sort      by key, stable, value bits + 1 (the sentinel sorts last)
//   runs      a flag where the key changes, exclusive scan = run number, run starts, run lengths (their padded sum tells the
//             host how many postings to allocate)
//   slots     one hash insert per RUN (distinct keys only, no contention on a slot), cnt[slot] = the run's length, then the
//             exclusive scan of the padded counts over the SLOTS: keys / cnt / off exactly as the atomic build leaves them —
//             everything downstream (rows, export, the query tables, the scoring kernels' partitions) is unchanged
//   place     posting i of run r goes to post[off[slot(r)] + i - start(r)]
//
// Needs value bits < 32 (k <= 15: the sentinel); the atomic build stays for k = 16 and above and behind IOC_BUILD_SORT=0.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ioc_kernels.h"

#define BS_BLOCK 256
#define BS_EMPTY 0xFFFFFFFFu

namespace {

__device__ __forceinline__ uint32_t bs_hash_slot(uint32_t v, uint32_t shift) { return (v * 0x9E3779B1u) >> shift; }  // (= hash_slot of ioc_kernels.hip)

// the dedicated slot `cap` belongs to the value 0xFFFFFFFF (the empty marker of the table), as in hash_insert
__device__ __forceinline__ uint32_t bs_hash_insert(uint32_t* __restrict__ keys, uint32_t cap, uint32_t shift, uint32_t v)
{
    if (v == BS_EMPTY) return cap;
    uint32_t h = bs_hash_slot(v, shift);
    for (uint32_t step = 0; step < cap; ++step) {
        const uint32_t k = keys[h];
        if (k == v) return h;
        if (k == BS_EMPTY) {
            const uint32_t old = atomicCAS(&keys[h], BS_EMPTY, v);
            if (old == BS_EMPTY || old == v) return h;
        }
        h = (h + 1) & (cap - 1);
    }
    return cap + 1;
}

template <typename PT>
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_pairs_queries(int n, uint32_t L, const int64_t* __restrict__ doff, const uint32_t* __restrict__ dcount, const uint32_t* __restrict__ dvals,
                   int64_t base, uint32_t sentinel, uint32_t* __restrict__ pk, PT* __restrict__ pv)
{
    const int j = blockIdx.x;
    if (j >= n) return;
    const int64_t b = doff[j];
    const uint32_t cap = uint32_t(doff[j + 1] - b), m = dcount[j];
    for (uint32_t d = threadIdx.x; d < cap; d += BS_BLOCK) {
        pk[base + b + d] = d < m ? dvals[b + d] : sentinel;
        pv[base + b + d] = PT(L + uint32_t(j));
    }
}

template <typename PT>
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_pairs_left(int64_t nkeys, const uint32_t* __restrict__ lkeys, const int64_t* __restrict__ loffs, const uint32_t* __restrict__ lpost,
                uint32_t* __restrict__ pk, PT* __restrict__ pv)
{
    // one wave per key
    const int64_t i = (int64_t(blockIdx.x) * BS_BLOCK + threadIdx.x) >> 6;
    if (i >= nkeys) return;
    const uint32_t key = lkeys[i];
    for (int64_t x = loffs[i] + (threadIdx.x & 63); x < loffs[i + 1]; x += 64) {
        pk[x] = key;
        pv[x] = PT(lpost[x]);
    }
}

// ---- runs of equal keys: run number of every pair, run starts, totals ------------------------------------------------------
// A pair opens a run if its key is a real one (below the sentinel) and differs from the key before it.  rid[i] = the number of
// runs opened BEFORE pair i (an exclusive scan of those flags; the run of pair i is rid[i] if it opens one, rid[i] - 1 otherwise),
// run_start[r] = the first pair of run r, run_start[R] = ctl[0] = the number of real pairs, ctl[1] = R.  Three launches — count
// per block, scan of the blocks' counts, number — that read the sorted keys twice and write rid once (rocPRIM's one-pass scan
// over a flag iterator followed by a kernel for the starts took 69 + 33 us for the bench batch's 12.3 M pairs; this 8 + 4 + 29 us).
constexpr int RUN_IPT = 8;                       // consecutive pairs per thread
constexpr int RUN_ELEMS = BS_BLOCK * RUN_IPT;    // ... per block

// exclusive prefix sum of v over the block's 256 threads; total = the block's sum
__device__ __forceinline__ uint32_t bs_block_scan(uint32_t v, uint32_t& total, uint32_t* sh)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= uint32_t(o)) inc += t;
    }
    if (lane == 63u) sh[wv] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < BS_BLOCK / 64; ++w) {
        const uint32_t x = sh[w];
        if (w < wv) before += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return before + inc - v;
}

// keys[0 .. RUN_IPT + 1]: the key before the thread's first pair (a value no key has if there is none), its pairs (the sentinel
// past the end), the key after them
__device__ __forceinline__ void bs_load_keys(const uint32_t* __restrict__ pk, int64_t P, int64_t i0, uint32_t sentinel, uint32_t (&keys)[RUN_IPT + 2])
{
    keys[0] = (i0 > 0 && i0 <= P) ? pk[i0 - 1] : 0xFFFFFFFFu;
    if (i0 + RUN_IPT <= P) {
        const uint4 a = *reinterpret_cast<const uint4*>(pk + i0), b = *reinterpret_cast<const uint4*>(pk + i0 + 4);  // (i0 is a multiple of 8, the array 16-byte aligned)
        keys[1] = a.x, keys[2] = a.y, keys[3] = a.z, keys[4] = a.w, keys[5] = b.x, keys[6] = b.y, keys[7] = b.z, keys[8] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < RUN_IPT; ++e) keys[1 + e] = i0 + e < P ? pk[i0 + e] : sentinel;
    }
    keys[RUN_IPT + 1] = i0 + RUN_IPT < P ? pk[i0 + RUN_IPT] : sentinel;
}

__global__ void __launch_bounds__(BS_BLOCK)
k_bs_run_count(int64_t P, const uint32_t* __restrict__ pk, uint32_t sentinel, uint32_t* __restrict__ bsum)
{
    __shared__ uint32_t sh[BS_BLOCK / 64];
    static_assert(RUN_IPT == 8, "bs_load_keys reads two 16-byte vectors");
    const int64_t i0 = int64_t(blockIdx.x) * RUN_ELEMS + int64_t(threadIdx.x) * RUN_IPT;
    uint32_t keys[RUN_IPT + 2];
    bs_load_keys(pk, P, i0, sentinel, keys);
    uint32_t n = 0;
#pragma unroll
    for (int e = 0; e < RUN_IPT; ++e) n += (keys[1 + e] < sentinel && keys[e] != keys[1 + e]) ? 1u : 0u;
    uint32_t tot;
    (void)bs_block_scan(n, tot, sh);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one block: the blocks' counts -> the number of runs opened before each block
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_run_sums(uint32_t* __restrict__ bsum, int64_t nb)
{
    __shared__ uint32_t sh[BS_BLOCK / 64];
    uint32_t carry = 0;
    for (int64_t c = 0; c < nb; c += RUN_ELEMS) {
        const int64_t i0 = c + int64_t(threadIdx.x) * RUN_IPT;
        uint32_t v[RUN_IPT], s = 0;
#pragma unroll
        for (int e = 0; e < RUN_IPT; ++e) {
            v[e] = i0 + e < nb ? bsum[i0 + e] : 0u;
            s += v[e];
        }
        uint32_t tot;
        uint32_t ex = bs_block_scan(s, tot, sh) + carry;
#pragma unroll
        for (int e = 0; e < RUN_IPT; ++e) {
            if (i0 + e < nb) bsum[i0 + e] = ex;
            ex += v[e];
        }
        carry += tot;
    }
}

__global__ void __launch_bounds__(BS_BLOCK)
k_bs_runs(int64_t P, const uint32_t* __restrict__ pk, uint32_t sentinel, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ rid,
          uint32_t* __restrict__ ctl, uint32_t* __restrict__ run_start)
{
    __shared__ uint32_t sh[BS_BLOCK / 64];
    const int64_t i0 = int64_t(blockIdx.x) * RUN_ELEMS + int64_t(threadIdx.x) * RUN_IPT;
    uint32_t keys[RUN_IPT + 2];
    bs_load_keys(pk, P, i0, sentinel, keys);
    uint32_t n = 0;
    bool first[RUN_IPT];
#pragma unroll
    for (int e = 0; e < RUN_IPT; ++e) {
        first[e] = keys[1 + e] < sentinel && keys[e] != keys[1 + e];
        n += first[e] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t ex = bs_block_scan(n, tot, sh) + bsum[blockIdx.x];
    uint32_t out[RUN_IPT];
#pragma unroll
    for (int e = 0; e < RUN_IPT; ++e) {
        out[e] = ex;
        if (first[e]) run_start[ex] = uint32_t(i0 + e);
        ex += first[e] ? 1u : 0u;
        // the last real pair (the keys are sorted: the sentinels, if any, follow it)
        if (i0 + e < P && keys[1 + e] < sentinel && (i0 + e + 1 == P || keys[2 + e] >= sentinel)) {
            run_start[ex] = uint32_t(i0 + e + 1);
            ctl[0] = uint32_t(i0 + e + 1);
            ctl[1] = ex;
        }
    }
    if (i0 + RUN_IPT <= P) {
        *reinterpret_cast<uint4*>(rid + i0) = uint4{out[0], out[1], out[2], out[3]};
        *reinterpret_cast<uint4*>(rid + i0 + 4) = uint4{out[4], out[5], out[6], out[7]};
    } else {
#pragma unroll
        for (int e = 0; e < RUN_IPT; ++e)
            if (i0 + e < P) rid[i0 + e] = out[e];
    }
}

// one thread per run: its key's slot (inserted here: distinct keys only) and the list's length.  The lists are laid out in SLOT
// order afterwards (an exclusive scan of the padded counts over the slots, as the atomic build does): the scoring kernels cut
// the table into 8 partitions by the top slot bits and want a partition's postings contiguous.
template <typename PT>
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_slots(uint32_t R, const uint32_t* __restrict__ pk, const PT* __restrict__ pv, const uint32_t* __restrict__ run_start,
           uint32_t* __restrict__ run_slot, uint32_t* __restrict__ keys, uint32_t cap, uint32_t shift, uint32_t* __restrict__ cnt,
           uint2* __restrict__ qinfo, Epochs E, uint32_t* __restrict__ err)
{
    const uint32_t r = blockIdx.x * BS_BLOCK + threadIdx.x;
    if (r >= R) return;
    const uint32_t slot = bs_hash_insert(keys, cap, shift, pk[run_start[r]]);
    if (slot > cap) {
        atomicAdd(err, 1u);
        return;
    }
    const uint32_t c = run_start[r + 1] - run_start[r];
    cnt[slot] = c;
    run_slot[r] = slot;
    // the row's epoch cuts (index_lookup): per boundary the number of entries below it, in units of 8 postings — the run is
    // ascending, a binary search per boundary
    uint2 info = make_uint2(c, 0x80000000u);
    if (c < IOC_EPOCH_LONG) {
        const PT* p = pv + run_start[r];
        uint32_t b[IOC_EPOCHS];
#pragma unroll
        for (int i = 0; i < IOC_EPOCHS; ++i) {
            uint32_t lo = 0, hi = c;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (uint32_t(p[mid]) < E.e[i]) lo = mid + 1; else hi = mid;
            }
            b[i] = (lo + 7u) >> 3;
        }
        info.x = c | (b[0] << 10) | (b[1] << 17) | (b[2] << 24);
        info.y = b[3] | (b[4] << 7) | (b[5] << 14) | (b[6] << 21);
    }
    qinfo[slot] = info;
}

template <typename PT>
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_place(uint32_t n_real, const uint32_t* __restrict__ pk, const PT* __restrict__ pv, const uint32_t* __restrict__ rid,
           const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ run_slot, const uint32_t* __restrict__ off, PT* __restrict__ post)
{
    const uint32_t i = blockIdx.x * BS_BLOCK + threadIdx.x;
    if (i >= n_real) return;
    const uint32_t r = (i == 0 || pk[i - 1] != pk[i]) ? rid[i] : rid[i] - 1u;
    post[off[run_slot[r]] + (i - run_start[r])] = pv[i];
}

}  // namespace

size_t iock_build_sort_temp_bytes(int64_t P, int post16, int value_bits)
{
    size_t tmp = 0;
    uint32_t* k = nullptr;
    if (post16) {
        uint16_t* v = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, tmp, k, k, v, v, size_t(P), 0, unsigned(value_bits + 1));
    } else {
        uint32_t* v = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, tmp, k, k, v, v, size_t(P), 0, unsigned(value_bits + 1));
    }
    const size_t tmp2 = (size_t(P) / RUN_ELEMS + 2) * 4;  // the run numbering's counts per block (after the sort: the same scratch)
    return (tmp > tmp2 ? tmp : tmp2) + 256;
}

// phase 1: pairs, sort, runs.  Afterwards ctl[0] = real pairs, ctl[1] = runs (= distinct keys).
hipError_t iock_build_sort_phase1(hipStream_t st, const IocBuildSort* a)
{
    const int64_t P = a->P;
    if (P <= 0) return hipSuccess;
    const uint32_t sentinel = 1u << a->value_bits;
    hipError_t e = hipMemsetAsync(a->ctl, 0, 16, st);
    if (e != hipSuccess) return e;
    if (a->n_left_keys > 0) {
        const int64_t threads = a->n_left_keys * 64;
        const dim3 grid(unsigned((threads + BS_BLOCK - 1) / BS_BLOCK));
        if (a->post16)
            hipLaunchKernelGGL(k_bs_pairs_left<uint16_t>, grid, dim3(BS_BLOCK), 0, st, a->n_left_keys, a->lkeys, a->loffs, a->lpost, a->pk_in,
                               static_cast<uint16_t*>(a->pv_in));
        else
            hipLaunchKernelGGL(k_bs_pairs_left<uint32_t>, grid, dim3(BS_BLOCK), 0, st, a->n_left_keys, a->lkeys, a->loffs, a->lpost, a->pk_in,
                               static_cast<uint32_t*>(a->pv_in));
    }
    if (a->n > 0 && !a->pairs_done) {
        if (a->post16)
            hipLaunchKernelGGL(k_bs_pairs_queries<uint16_t>, dim3(a->n), dim3(BS_BLOCK), 0, st, a->n, a->L, a->doff, a->dcount, a->dvals, a->n_left_post,
                               sentinel, a->pk_in, static_cast<uint16_t*>(a->pv_in));
        else
            hipLaunchKernelGGL(k_bs_pairs_queries<uint32_t>, dim3(a->n), dim3(BS_BLOCK), 0, st, a->n, a->L, a->doff, a->dcount, a->dvals, a->n_left_post,
                               sentinel, a->pk_in, static_cast<uint32_t*>(a->pv_in));
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tmp = a->temp_bytes;
    if (a->post16)
        e = rocprim::radix_sort_pairs(a->temp, tmp, a->pk_in, a->pk_out, static_cast<uint16_t*>(a->pv_in), static_cast<uint16_t*>(a->pv_out), size_t(P), 0,
                                      unsigned(a->value_bits + 1), st);
    else
        e = rocprim::radix_sort_pairs(a->temp, tmp, a->pk_in, a->pk_out, static_cast<uint32_t*>(a->pv_in), static_cast<uint32_t*>(a->pv_out), size_t(P), 0,
                                      unsigned(a->value_bits + 1), st);
    if (e != hipSuccess) return e;
    // run numbers, run starts, totals
    const int64_t nb = (P + RUN_ELEMS - 1) / RUN_ELEMS;
    if ((size_t(nb) + 1) * 4 > a->temp_bytes) return hipErrorInvalidValue;
    uint32_t* bsum = static_cast<uint32_t*>(a->temp);
    hipLaunchKernelGGL(k_bs_run_count, dim3(unsigned(nb)), dim3(BS_BLOCK), 0, st, P, a->pk_out, sentinel, bsum);
    hipLaunchKernelGGL(k_bs_run_sums, dim3(1), dim3(BS_BLOCK), 0, st, bsum, nb);
    hipLaunchKernelGGL(k_bs_runs, dim3(unsigned(nb)), dim3(BS_BLOCK), 0, st, P, a->pk_out, sentinel, bsum, a->rid, a->ctl, a->run_start);
    return hipGetLastError();
}

// phase 2 (the host knows R, the real pairs and the table's capacity now): slots and postings
hipError_t iock_build_sort_phase2(hipStream_t st, const IocBuildSort* a, uint32_t R, uint32_t n_real, uint32_t* keys, uint32_t cap, uint32_t shift,
                                  uint32_t* cnt, uint32_t* off, void* post, uint32_t* qinfo, uint32_t n_targets, uint32_t* err)
{
    if (R == 0) return hipSuccess;
    // (the scan's contract: ceil(n / 1024) + 1 words of scratch for n = cap + 1 slots; the caller sized it for the largest table)
    if ((size_t(cap) + 1 + 1023) / 1024 + 1 > a->scan_words) return hipErrorInvalidValue;
    const Epochs E = iock_epoch_bounds(a->L, n_targets);
    const dim3 gr((R + BS_BLOCK - 1) / BS_BLOCK);
    if (a->post16)
        hipLaunchKernelGGL(k_bs_slots<uint16_t>, gr, dim3(BS_BLOCK), 0, st, R, a->pk_out, static_cast<const uint16_t*>(a->pv_out), a->run_start,
                           a->run_slot, keys, cap, shift, cnt, reinterpret_cast<uint2*>(qinfo), E, err);
    else
        hipLaunchKernelGGL(k_bs_slots<uint32_t>, gr, dim3(BS_BLOCK), 0, st, R, a->pk_out, static_cast<const uint32_t*>(a->pv_out), a->run_start,
                           a->run_slot, keys, cap, shift, cnt, reinterpret_cast<uint2*>(qinfo), E, err);
    hipError_t e = iock_exclusive_scan(st, cnt, int64_t(cap) + 1, off, a->scan_scratch, a->pad_mask);
    if (e != hipSuccess) return e;
    const dim3 g((n_real + BS_BLOCK - 1) / BS_BLOCK);
    if (a->post16)
        hipLaunchKernelGGL(k_bs_place<uint16_t>, g, dim3(BS_BLOCK), 0, st, n_real, a->pk_out, static_cast<const uint16_t*>(a->pv_out), a->rid, a->run_start, a->run_slot,
                           off, static_cast<uint16_t*>(post));
    else
        hipLaunchKernelGGL(k_bs_place<uint32_t>, g, dim3(BS_BLOCK), 0, st, n_real, a->pk_out, static_cast<const uint32_t*>(a->pv_out), a->rid, a->run_start, a->run_slot,
                           off, static_cast<uint32_t*>(post));
    return hipGetLastError();
}

// (ioc_ctx_prewarm: makes the runtime load this file's code object now instead of at its first launch)
extern "C" hipError_t iock_warm_build_sort()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_bs_pairs_queries<uint16_t>));
}
