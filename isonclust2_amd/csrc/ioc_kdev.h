// ioc_kdev.h — what the kernel files of the assignment path share on the device side: launch geometry, tuning macros, the
// index row lookup, wave / workgroup scans, the query -> workgroup mapping of the sharded merge.  Included by ioc_kernels.hip
// (index build), ioc_score.hip (scoring, hit tables) and ioc_resolve.hip (gap bounds, decide / evaluate / pick) only.
#pragma once
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


#define CK(x)                     \
    do {                          \
        hipError_t e_ = (x);      \
        if (e_ != hipSuccess) return e_; \
    } while (0)

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

}  // namespace
