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

// ---- queries beyond the in-LDS sort of k_distinct_radix (more than IOC_DISTINCT_LDS_MAX forward minimizers: reads of ~35 kb and
// up, chimeric ultra-long ones of hundreds of kb) --------------------------------------------------------------------------------
// Their values are gathered side by side, sorted segment by segment with rocPRIM (library code for a rare case: the reference's
// std::set has no limit, minimizer.cpp:31-42, so neither has this path), and written out like the short queries': distinct values,
// their count, and — sorted index build — the (value, target) pairs with the sentinel behind them.
namespace {

__global__ void __launch_bounds__(BLK)
k_long_gather(const int32_t* __restrict__ qid, const int64_t* __restrict__ off_fwd, const unsigned long long* __restrict__ seg, const uint32_t* __restrict__ mins,
              uint32_t* __restrict__ out)
{
    const int j = qid[blockIdx.x];
    const int64_t b = off_fwd[j];
    const uint32_t m = uint32_t(off_fwd[j + 1] - b);
    uint32_t* o = out + seg[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < m; i += BLK) o[i] = mins[b + i];
}

__global__ void __launch_bounds__(BLK)
k_long_unique(const int32_t* __restrict__ qid, const unsigned long long* __restrict__ seg, const uint32_t* __restrict__ sorted, const int64_t* __restrict__ doff,
              uint32_t* __restrict__ dvals, uint32_t* __restrict__ dcount, uint32_t* __restrict__ pk, void* __restrict__ pv, int pv16, uint32_t target0,
              uint32_t sentinel)
{
    __shared__ uint32_t wsum[BLK / 64];
    __shared__ uint32_t s_base;
    const int j = qid[blockIdx.x];
    const uint32_t* s = sorted + seg[blockIdx.x];
    const uint32_t m = uint32_t(seg[blockIdx.x + 1] - seg[blockIdx.x]);
    uint32_t* out = dvals + doff[j];
    uint32_t* pko = pk ? pk + doff[j] : nullptr;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t c = 0; c < m; c += BLK) {
        const uint32_t i = c + threadIdx.x;
        const bool flag = i < m && (i == 0 || s[i] != s[i - 1]);
        const unsigned long long bm = __ballot(flag);
        if (lane == 0) wsum[wave] = uint32_t(__popcll(bm));
        __syncthreads();
        uint32_t before = s_base;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (flag) {
            const uint32_t at = before + uint32_t(__popcll(bm & ((1ull << lane) - 1ull)));
            out[at] = s[i];
            if (pko) pko[at] = s[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < BLK / 64; ++w) t += wsum[w];
            s_base += t;
        }
        __syncthreads();
    }
    const uint32_t base = s_base;
    if (threadIdx.x == 0) dcount[j] = base;
    if (pko) {
        for (uint32_t d = base + threadIdx.x; d < m; d += BLK) pko[d] = sentinel;
        const uint32_t t = target0 + uint32_t(j);
        if (pv16) {
            uint16_t* o = static_cast<uint16_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += BLK) o[d] = uint16_t(t);
        } else {
            uint32_t* o = static_cast<uint32_t*>(pv) + doff[j];
            for (uint32_t d = threadIdx.x; d < m; d += BLK) o[d] = t;
        }
    }
}

}  // namespace

size_t iock_distinct_long_temp(size_t total, uint32_t nlong, int value_bits)
{
    size_t a = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, unsigned(total), nlong, (const unsigned long long*)nullptr,
                                             (const unsigned long long*)nullptr, 0u, unsigned(value_bits), hipStream_t(nullptr));
    return a + 256;
}

// work = [total words: gathered][total words: sorted][temp]; d_qid [nlong], d_seg [nlong + 1] (offsets of the segments in the gathered array)
hipError_t iock_distinct_long(hipStream_t st, uint32_t nlong, size_t total, const int32_t* d_qid, const unsigned long long* d_seg, const int64_t* off_fwd,
                              const uint32_t* mins, const int64_t* doff, uint32_t* dvals, uint32_t* dcount, int value_bits, uint32_t* work, void* temp,
                              size_t temp_bytes, uint32_t* pk, void* pv, int pv16, uint32_t target0, uint32_t sentinel)
{
    if (!nlong) return hipSuccess;
    uint32_t* in = work;
    uint32_t* out = work + total;
    hipLaunchKernelGGL(k_long_gather, dim3(nlong), dim3(BLK), 0, st, d_qid, off_fwd, d_seg, mins, in);
    size_t tb = temp_bytes;
    const int bits = value_bits < 1 || value_bits > 32 ? 32 : value_bits;
    hipError_t e = rocprim::segmented_radix_sort_keys(temp, tb, in, out, unsigned(total), nlong, d_seg, d_seg + 1, 0u, unsigned(bits), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_long_unique, dim3(nlong), dim3(BLK), 0, st, d_qid, d_seg, out, doff, dvals, dcount, pk, pv, pv16, target0, sentinel);
    return hipGetLastError();
}
