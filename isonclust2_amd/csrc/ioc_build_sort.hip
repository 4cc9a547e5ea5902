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

// flag = 1 at the first pair of every run of real keys; ctl[0] = number of real pairs
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_flags(int64_t P, const uint32_t* __restrict__ pk, uint32_t sentinel, uint32_t* __restrict__ flags, uint32_t* __restrict__ ctl)
{
    const int64_t i = int64_t(blockIdx.x) * BS_BLOCK + threadIdx.x;
    if (i >= P) return;
    const uint32_t key = pk[i];
    const bool real = key < sentinel;
    flags[i] = (real && (i == 0 || pk[i - 1] != key)) ? 1u : 0u;
    if (real && (i + 1 == P || pk[i + 1] >= sentinel)) ctl[0] = uint32_t(i + 1);
}

// run_start[r] = index of run r's first pair; run_start[R] = number of real pairs
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_starts(int64_t P, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ rid, const uint32_t* __restrict__ ctl,
            uint32_t* __restrict__ run_start)
{
    const int64_t i = int64_t(blockIdx.x) * BS_BLOCK + threadIdx.x;
    if (i >= P) return;
    if (flags[i]) run_start[rid[i]] = uint32_t(i);
    if (i == 0) run_start[rid[P]] = ctl[0];  // (rid[P] = R: the scan's total)
}

__global__ void __launch_bounds__(BS_BLOCK)
k_bs_lens(int64_t P, const uint32_t* __restrict__ rid, const uint32_t* __restrict__ run_start, uint32_t* __restrict__ lens)
{
    const int64_t r = int64_t(blockIdx.x) * BS_BLOCK + threadIdx.x;
    if (r >= P) return;
    lens[r] = r < int64_t(rid[P]) ? run_start[r + 1] - run_start[r] : 0u;
}

// one thread per run: its key's slot (inserted here: distinct keys only) and the list's length.  The lists are laid out in SLOT
// order afterwards (an exclusive scan of the padded counts over the slots, as the atomic build does): the scoring kernels cut
// the table into 8 partitions by the top slot bits and want a partition's postings contiguous.
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_slots(uint32_t R, const uint32_t* __restrict__ pk, const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ lens,
           uint32_t* __restrict__ run_slot, uint32_t* __restrict__ keys, uint32_t cap, uint32_t shift, uint32_t* __restrict__ cnt,
           uint32_t* __restrict__ err)
{
    const uint32_t r = blockIdx.x * BS_BLOCK + threadIdx.x;
    if (r >= R) return;
    const uint32_t slot = bs_hash_insert(keys, cap, shift, pk[run_start[r]]);
    if (slot > cap) {
        atomicAdd(err, 1u);
        return;
    }
    cnt[slot] = lens[r];
    run_slot[r] = slot;
}

template <typename PT>
__global__ void __launch_bounds__(BS_BLOCK)
k_bs_place(uint32_t n_real, const PT* __restrict__ pv, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ rid,
           const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ run_slot, const uint32_t* __restrict__ off, PT* __restrict__ post)
{
    const uint32_t i = blockIdx.x * BS_BLOCK + threadIdx.x;
    if (i >= n_real) return;
    const uint32_t r = flags[i] ? rid[i] : rid[i] - 1u;
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
    return tmp + 256;
}

// phase 1: pairs, sort, runs, padded offsets.  Afterwards ctl[0] = real pairs, rid[P] = runs, roff[P] = padded postings.
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
    if (a->n > 0) {
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
    const dim3 gp(unsigned((P + BS_BLOCK - 1) / BS_BLOCK));
    uint32_t* flags = a->pk_in;  // (free again)
    hipLaunchKernelGGL(k_bs_flags, gp, dim3(BS_BLOCK), 0, st, P, a->pk_out, sentinel, flags, a->ctl);
    e = iock_exclusive_scan(st, flags, P, a->rid, a->scan_scratch, 0u);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bs_starts, gp, dim3(BS_BLOCK), 0, st, P, flags, a->rid, a->ctl, a->run_start);
    hipLaunchKernelGGL(k_bs_lens, gp, dim3(BS_BLOCK), 0, st, P, a->rid, a->run_start, a->lens);
    e = iock_exclusive_scan(st, a->lens, P, a->roff, a->scan_scratch, a->pad_mask);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// phase 2 (the host knows R, the real pairs and the table's capacity now): slots and postings
hipError_t iock_build_sort_phase2(hipStream_t st, const IocBuildSort* a, uint32_t R, uint32_t n_real, uint32_t* keys, uint32_t cap, uint32_t shift,
                                  uint32_t* cnt, uint32_t* off, void* post, uint32_t* err)
{
    if (R == 0) return hipSuccess;
    uint32_t* run_slot = a->roff;  // (the padded offsets in run order have told the host the total: the array is free)
    hipLaunchKernelGGL(k_bs_slots, dim3((R + BS_BLOCK - 1) / BS_BLOCK), dim3(BS_BLOCK), 0, st, R, a->pk_out, a->run_start, a->lens, run_slot, keys, cap, shift, cnt,
                       err);
    hipError_t e = iock_exclusive_scan(st, cnt, int64_t(cap) + 1, off, a->scan_scratch, a->pad_mask);
    if (e != hipSuccess) return e;
    const uint32_t* flags = a->pk_in;
    const dim3 g((n_real + BS_BLOCK - 1) / BS_BLOCK);
    if (a->post16)
        hipLaunchKernelGGL(k_bs_place<uint16_t>, g, dim3(BS_BLOCK), 0, st, n_real, static_cast<const uint16_t*>(a->pv_out), flags, a->rid, a->run_start, run_slot,
                           off, static_cast<uint16_t*>(post));
    else
        hipLaunchKernelGGL(k_bs_place<uint32_t>, g, dim3(BS_BLOCK), 0, st, n_real, static_cast<const uint32_t*>(a->pv_out), flags, a->rid, a->run_start, run_slot,
                           off, static_cast<uint32_t*>(post));
    return hipGetLastError();
}
