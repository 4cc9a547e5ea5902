// ioc_sort.hip — ordering of the MinDB's keys for ioc_index_export on the device.
// The reference's MinDB is an unordered_map (src/minimizer.h): the exported CSR lists the keys in ascending order.  The
// hash table holds them in slot order; here the slots are sorted by (list is empty, key) with rocPRIM's radix sort
// (library code for a utility step: 1 M slots in ~0.1 ms; the host's std::sort took 9 of the export's 11 ms), the kept
// counts are scanned in that order, and every slot learns where its list goes.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>

#include "ioc_kernels.h"

namespace {

constexpr int BLK = 256;

__global__ void __launch_bounds__(BLK)
k_order_keys(uint32_t nslots, uint32_t cap, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ cnt,
             unsigned long long* __restrict__ skey, uint32_t* __restrict__ sval, uint32_t* __restrict__ n_rows)
{
    const uint32_t s = blockIdx.x * BLK + threadIdx.x;
    const bool in = s < nslots;
    const bool kept = in && cnt[s] != 0;
    if (in) {
        const uint32_t k = s == cap ? 0xFFFFFFFFu : keys[s];  // (slot `cap` is the list of the value 0xFFFFFFFF)
        skey[s] = (kept ? 0ull : 1ull << 32) | k;
        sval[s] = s;
    }
    const unsigned long long b = __ballot(kept);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_rows, uint32_t(__popcll(b)));
}

__global__ void __launch_bounds__(BLK)
k_sorted_counts(uint32_t nslots, const uint32_t* __restrict__ sval, const uint32_t* __restrict__ cnt, unsigned long long* __restrict__ scnt)
{
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i < nslots) scnt[i] = cnt[sval[i]];
    if (i == nslots) scnt[i] = 0;  // (the scan's last output is the total)
}

__global__ void __launch_bounds__(BLK)
k_slot_offsets(uint32_t nslots, const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sval,
               const unsigned long long* __restrict__ soff, uint32_t* __restrict__ okeys, int64_t* __restrict__ slot_off)
{
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nslots) return;
    okeys[i] = uint32_t(skey[i]);
    slot_off[sval[i]] = int64_t(soff[i]);
}

}  // namespace

size_t iock_export_order_temp(uint32_t nslots)
{
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, size_t(nslots), 0, 33, hipStream_t(nullptr));
    (void)rocprim::exclusive_scan(nullptr, b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, size_t(nslots) + 1,
                                  rocprim::plus<unsigned long long>(), hipStream_t(nullptr));
    return (a > b ? a : b) + 256;
}

// work: 2 x nslots u64 (keys in / out) + 2 x nslots u32 (slots in / out) + (nslots + 1) u64 (counts, scanned in place into
// `soff`) are carved from `work` by the caller: see ioc_capi.cpp
hipError_t iock_export_order(hipStream_t st, uint32_t nslots, uint32_t cap, const uint32_t* keys, const uint32_t* cnt,
                             unsigned long long* k0, unsigned long long* k1, uint32_t* v0, uint32_t* v1, unsigned long long* scnt,
                             unsigned long long* soff, void* temp, size_t temp_bytes, uint32_t* n_rows, uint32_t* okeys, int64_t* slot_off)
{
    hipError_t e;
    if ((e = hipMemsetAsync(n_rows, 0, 4, st)) != hipSuccess) return e;
    const unsigned nb = (nslots + 1 + BLK - 1) / BLK;
    hipLaunchKernelGGL(k_order_keys, dim3(nb), dim3(BLK), 0, st, nslots, cap, keys, cnt, k0, v0, n_rows);
    size_t tb = temp_bytes;
    if ((e = rocprim::radix_sort_pairs(temp, tb, k0, k1, v0, v1, size_t(nslots), 0, 33, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_sorted_counts, dim3(nb), dim3(BLK), 0, st, nslots, v1, cnt, scnt);
    tb = temp_bytes;
    if ((e = rocprim::exclusive_scan(temp, tb, scnt, soff, 0ull, size_t(nslots) + 1, rocprim::plus<unsigned long long>(), st)) != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_slot_offsets, dim3(nb), dim3(BLK), 0, st, nslots, k1, v1, soff, okeys, slot_off);
    return hipGetLastError();
}

// (ioc_ctx_prewarm: makes the runtime load this file's code object now instead of at its first launch)
extern "C" hipError_t iock_warm_sort()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_order_keys));
}
