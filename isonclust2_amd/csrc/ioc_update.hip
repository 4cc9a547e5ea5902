// ioc_update.hip — UpdateMinDB on the device (src/minimizer.cpp:124-160, called at src/cluster.cpp:296 after a
// representative's consensus changed): cluster `cls` leaves the posting lists of the values that only the
// old representative had and enters — at its sorted place — the lists of the values only the new one has.
// Lists that become empty stay in the index as keys (the reference's erase is commented out, :150-152), and
// a value seen for the first time opens a new key.
//
// The persisted MinDB lives in HBM as CSR (sorted keys, offsets, ascending posting lists; ioc_left_load), so
// an update is one streaming rewrite of it: classify every key against the two small difference sets,
// place old and new keys in the merged order, exclusive scan of the new list lengths, copy every list
// with its one-element edit.  HBM-bound: 4 B per posting read + 4 B written, 12 B per key.
// The difference sets themselves (a few thousand values of ONE representative) are formed on the host from
// the caller's host arrays; the per-cluster value sets used by getMappedRatio (transposed MinDB) get the
// cluster's segment replaced in the same pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ioc_internal.h"
#include "ioc_kernels.h"

namespace {

__device__ __forceinline__ int64_t lower_bound_dev(const uint32_t* a, int64_t n, uint32_t v)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// kind per old key: 0 untouched, 1 in toDel, 2 in toIns; hit_* mark the difference-set entries that already
// are keys of the index
__global__ void __launch_bounds__(256)
k_upd_classify(int64_t n_keys, const uint32_t* __restrict__ keys, const int64_t* __restrict__ offs,
               const uint32_t* __restrict__ post, uint32_t cls, const uint32_t* __restrict__ to_del, int64_t n_del,
               const uint32_t* __restrict__ to_ins, int64_t n_ins, uint8_t* __restrict__ kind, uint32_t* __restrict__ newlen,
               uint8_t* __restrict__ hit_del, uint8_t* __restrict__ hit_ins)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint32_t key = keys[i];
    const int64_t a = offs[i], b = offs[i + 1];
    uint32_t len = uint32_t(b - a);
    uint8_t kd = 0;
    const int64_t d = lower_bound_dev(to_del, n_del, key);
    if (d < n_del && to_del[d] == key) {
        kd = 1;
        hit_del[d] = 1;
        // std::set(mins).erase(best): the list is strictly ascending (ioc_left_load), so at most one entry goes
        const int64_t p = lower_bound_dev(post + a, b - a, cls);
        if (p < b - a && post[a + p] == cls) len -= 1;
    } else {
        const int64_t s = lower_bound_dev(to_ins, n_ins, key);
        if (s < n_ins && to_ins[s] == key) {
            kd = 2;
            hit_ins[s] = 1;
            len += 1;  // push_back(best) + sort: a duplicate if best was already there, as in the reference
        }
    }
    kind[i] = kd;
    newlen[i] = len;
}

// merged key order: old key i moves behind the absent (new) keys smaller than it, absent key a behind the
// old keys smaller than it
__global__ void __launch_bounds__(256)
k_upd_place(int64_t n_keys, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ newlen,
            int64_t n_abs, const uint32_t* __restrict__ abs_key, const uint8_t* __restrict__ abs_ins,
            uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_len, int64_t* __restrict__ out_src)
{
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n_keys) {
        const int64_t pos = t + lower_bound_dev(abs_key, n_abs, keys[t]);
        out_keys[pos] = keys[t];
        out_len[pos] = newlen[t];
        out_src[pos] = t;
    } else if (t < n_keys + n_abs) {
        const int64_t a = t - n_keys;
        const int64_t pos = a + lower_bound_dev(keys, n_keys, abs_key[a]);
        out_keys[pos] = abs_key[a];
        out_len[pos] = abs_ins[a] ? 1u : 0u;
        out_src[pos] = abs_ins[a] ? -2 : -1;
    }
}

// one wave per new key: copy its list with the edit
__global__ void __launch_bounds__(256)
k_upd_copy(int64_t n_new_keys, const int64_t* __restrict__ src, const uint32_t* __restrict__ new_off32,
           const int64_t* __restrict__ offs, const uint32_t* __restrict__ post, const uint8_t* __restrict__ kind,
           uint32_t cls, uint32_t* __restrict__ out_post, int64_t* __restrict__ out_offs, uint32_t total)
{
    const int64_t k = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (k > n_new_keys) return;
    if (k == n_new_keys) {
        if (lane == 0) out_offs[k] = int64_t(total);
        return;
    }
    const uint32_t o = new_off32[k];
    if (lane == 0) out_offs[k] = int64_t(o);
    const int64_t s = src[k];
    if (s < 0) {
        if (s == -2 && lane == 0) out_post[o] = cls;
        return;
    }
    const int64_t a = offs[s], len = offs[s + 1] - a;
    const uint8_t kd = kind[s];
    if (kd == 2) {  // insert at the sorted place: entries <= cls stay, larger ones move up
        const int64_t at = lower_bound_dev(post + a, len, cls + 1u);
        for (int64_t e = lane; e < len; e += 64) out_post[o + e + (e >= at ? 1 : 0)] = post[a + e];
        if (lane == 0) out_post[o + at] = cls;
    } else if (kd == 1) {  // drop cls if present
        const int64_t at = lower_bound_dev(post + a, len, cls);
        const bool present = at < len && post[a + at] == cls;
        for (int64_t e = lane; e < len; e += 64) {
            if (present && e == at) continue;
            out_post[o + e - ((present && e > at) ? 1 : 0)] = post[a + e];
        }
    } else {
        for (int64_t e = lane; e < len; e += 64) out_post[o + e] = post[a + e];
    }
}

// per-cluster value sets (transposed MinDB): the segment of `cls` is replaced by its new sorted set
__global__ void __launch_bounds__(256)
k_upd_sets(int32_t L, int32_t cls, const int64_t* __restrict__ set_off, const uint32_t* __restrict__ set_val,
           int64_t old_total, const uint32_t* __restrict__ new_set, int64_t n_new, int64_t* __restrict__ out_off,
           uint32_t* __restrict__ out_val)
{
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t a = set_off[cls], b = set_off[cls + 1];
    const int64_t shift = n_new - (b - a);
    if (t <= L) out_off[t] = set_off[t] + (t > cls ? shift : 0);
    if (t < a)
        out_val[t] = set_val[t];
    else if (t >= b && t < old_total)
        out_val[t + shift] = set_val[t];
    if (t < n_new) out_val[a + t] = new_set[t];
}

struct Tmp {
    void* p = nullptr;
    ~Tmp()
    {
        if (p) (void)hipFree(p);
    }
    template <class T>
    T* as() const
    {
        return static_cast<T*>(p);
    }
};

}  // namespace

#define UCHK(c, call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

static int dev_alloc(ioc_ctx* c, Tmp& t, size_t bytes)
{
    const hipError_t e = hipMalloc(&t.p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        t.p = nullptr;
        return ioc_fail(c, IOC_ERR_CAPACITY, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    }
    ioc_poison(t.p, bytes ? bytes : 16);
    return IOC_OK;
}

extern "C" {

int ioc_index_update(ioc_ctx* c, int32_t cls, const uint32_t* old_min, int64_t n_old, const uint32_t* new_min,
                     int64_t n_new, uint8_t new_err_cell)
{
    if (!c || n_old < 0 || n_new < 0 || (n_old > 0 && !old_min) || (n_new > 0 && !new_min)) return IOC_ERR_ARG;
    UCHK(c, hipSetDevice(c->device));
    if (cls < 0 || cls >= c->L) return ioc_fail(c, IOC_ERR_ARG, "ioc_index_update: cluster id outside the left clusters");
    if (new_err_cell > 15) return ioc_fail(c, IOC_ERR_ARG, "ioc_index_update: err_cell outside 1..15 (0 = unchanged)");
    // ---- the two difference sets (std::set + set_difference of minimizer.cpp:127-143), on the host ----
    std::vector<uint32_t> olds(old_min, old_min + n_old), news(new_min, new_min + n_new), to_del, to_ins;
    std::sort(olds.begin(), olds.end());
    olds.erase(std::unique(olds.begin(), olds.end()), olds.end());
    std::sort(news.begin(), news.end());
    news.erase(std::unique(news.begin(), news.end()), news.end());
    std::set_difference(olds.begin(), olds.end(), news.begin(), news.end(), std::back_inserter(to_del));
    std::set_difference(news.begin(), news.end(), olds.begin(), olds.end(), std::back_inserter(to_ins));
    hipStream_t s = c->stream;
    // the old minimizers must be what the index holds for this cluster (always true in the reference: they were
    // added by AddMinimizers / a previous UpdateMinDB); anything else would silently corrupt the value sets
    {
        const int64_t a = c->h_lset_off[size_t(cls)], b = c->h_lset_off[size_t(cls) + 1];
        std::vector<uint32_t> cur(size_t(b - a));
        if (b > a) UCHK(c, hipMemcpyAsync(cur.data(), static_cast<uint32_t*>(c->b_lset_val.p) + a, size_t(b - a) * 4, hipMemcpyDeviceToHost, s));
        UCHK(c, hipStreamSynchronize(s));
        if (cur != olds)
            return ioc_fail(c, IOC_ERR_INPUT, "ioc_index_update: old minimizers differ from the cluster's values in the index");
    }
    if (new_err_cell) {  // the consensus also re-weights the representative's HPC error rate (consensus.cpp:56-58)
        UCHK(c, hipMemcpyAsync(static_cast<uint8_t*>(c->b_left_err.p) + cls, &new_err_cell, 1, hipMemcpyHostToDevice, s));
        UCHK(c, hipStreamSynchronize(s));
        c->built = c->scored = c->resolved = false;
    }
    const int64_t n_keys = c->n_left_keys, n_post = c->n_left_post;
    const int64_t n_del = int64_t(to_del.size()), n_ins = int64_t(to_ins.size());
    if (n_del == 0 && n_ins == 0) return IOC_OK;
    int r;
    Tmp d_del, d_ins, d_kind, d_newlen, d_hit;
    if ((r = dev_alloc(c, d_del, size_t(n_del) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_ins, size_t(n_ins) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_kind, size_t(n_keys))) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_newlen, size_t(n_keys) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_hit, size_t(n_del + n_ins))) != IOC_OK) return r;
    if (n_del) UCHK(c, hipMemcpyAsync(d_del.p, to_del.data(), size_t(n_del) * 4, hipMemcpyHostToDevice, s));
    if (n_ins) UCHK(c, hipMemcpyAsync(d_ins.p, to_ins.data(), size_t(n_ins) * 4, hipMemcpyHostToDevice, s));
    UCHK(c, hipMemsetAsync(d_hit.p, 0, size_t(n_del + n_ins) ? size_t(n_del + n_ins) : 1, s));
    if (n_keys > 0) {
        hipLaunchKernelGGL(k_upd_classify, dim3(uint32_t((n_keys + 255) / 256)), dim3(256), 0, s, n_keys,
                           static_cast<const uint32_t*>(c->b_lkeys.p), static_cast<const int64_t*>(c->b_loffs.p),
                           static_cast<const uint32_t*>(c->b_lpost.p), uint32_t(cls), d_del.as<uint32_t>(), n_del,
                           d_ins.as<uint32_t>(), n_ins, d_kind.as<uint8_t>(), d_newlen.as<uint32_t>(), d_hit.as<uint8_t>(),
                           d_hit.as<uint8_t>() + n_del);
        UCHK(c, hipGetLastError());
    }
    std::vector<uint8_t> hit(size_t(n_del + n_ins) + 1, 0);
    UCHK(c, hipMemcpyAsync(hit.data(), d_hit.p, size_t(n_del + n_ins), hipMemcpyDeviceToHost, s));
    UCHK(c, hipStreamSynchronize(s));
    // difference-set values that are not keys yet: `db[m]` creates them (empty for toDel, [cls] for toIns)
    std::vector<uint32_t> abs_key;
    std::vector<uint8_t> abs_ins;
    {
        size_t i = 0, j = 0;
        while (i < size_t(n_del) || j < size_t(n_ins)) {  // both sorted and disjoint: merge
            const bool take_del = j >= size_t(n_ins) || (i < size_t(n_del) && to_del[i] < to_ins[j]);
            if (take_del) {
                if (!hit[i]) {
                    abs_key.push_back(to_del[i]);
                    abs_ins.push_back(0);
                }
                ++i;
            } else {
                if (!hit[size_t(n_del) + j]) {
                    abs_key.push_back(to_ins[j]);
                    abs_ins.push_back(1);
                }
                ++j;
            }
        }
    }
    const int64_t n_abs = int64_t(abs_key.size()), n_new_keys = n_keys + n_abs;
    if (n_post + n_ins >= (int64_t(1) << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 2^31 index postings");
    Tmp d_abs_key, d_abs_ins, d_len, d_src, d_off32, d_scan;
    DevBuf nk, no, np, nso, nsv;  // the new left state
    auto drop = [&]() {
        for (DevBuf* b : {&nk, &no, &np, &nso, &nsv})
            if (b->p) (void)hipFree(b->p);
    };
    auto grab = [&](DevBuf& b, size_t bytes) {
        b.cap = bytes ? bytes : 16;
        if (hipMalloc(&b.p, b.cap) != hipSuccess) return false;
        ioc_poison(b.p, b.cap);
        return true;
    };
    if ((r = dev_alloc(c, d_abs_key, size_t(n_abs) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_abs_ins, size_t(n_abs))) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_len, size_t(n_new_keys + 1) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_src, size_t(n_new_keys) * 8)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_off32, size_t(n_new_keys + 1) * 4)) != IOC_OK) return r;
    if ((r = dev_alloc(c, d_scan, size_t((n_new_keys + 1) / 256 + 1024) * 4 * 2)) != IOC_OK) return r;
    if (n_abs) {
        UCHK(c, hipMemcpyAsync(d_abs_key.p, abs_key.data(), size_t(n_abs) * 4, hipMemcpyHostToDevice, s));
        UCHK(c, hipMemcpyAsync(d_abs_ins.p, abs_ins.data(), size_t(n_abs), hipMemcpyHostToDevice, s));
    }
    const int64_t set_total = c->h_lset_off[size_t(c->L)];
    const int64_t new_set_total = set_total - int64_t(olds.size()) + int64_t(news.size());
    // postings: every toDel key that held cls loses one, every toIns key gains one (exact total: the scan)
    if (!grab(nk, size_t(n_new_keys) * 4) || !grab(no, size_t(n_new_keys + 1) * 8) ||
        !grab(np, size_t(n_post + n_ins) * 4) || !grab(nso, size_t(c->L + 1) * 8) ||
        !grab(nsv, size_t(new_set_total) * 4)) {
        drop();
        return ioc_fail(c, IOC_ERR_CAPACITY, "ioc_index_update: hipMalloc of the rewritten index failed");
    }
    hipLaunchKernelGGL(k_upd_place, dim3(uint32_t((n_new_keys + 255) / 256)), dim3(256), 0, s, n_keys,
                       static_cast<const uint32_t*>(c->b_lkeys.p), d_newlen.as<uint32_t>(), n_abs, d_abs_key.as<uint32_t>(),
                       d_abs_ins.as<uint8_t>(), static_cast<uint32_t*>(nk.p), d_len.as<uint32_t>(), d_src.as<int64_t>());
    if (hipGetLastError() != hipSuccess) {
        drop();
        return ioc_fail(c, IOC_ERR_HIP, "k_upd_place launch failed");
    }
    if (iock_exclusive_scan(s, d_len.as<uint32_t>(), n_new_keys, d_off32.as<uint32_t>(), d_scan.as<uint32_t>(), 0u) != hipSuccess) {
        drop();
        return ioc_fail(c, IOC_ERR_HIP, "exclusive scan of the new list lengths failed");
    }
    uint32_t total = 0;  // out[n] of the scan
    UCHK(c, hipMemcpyAsync(&total, d_off32.as<uint32_t>() + n_new_keys, 4, hipMemcpyDeviceToHost, s));
    UCHK(c, hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_upd_copy, dim3(uint32_t(((n_new_keys + 1) * 64 + 255) / 256)), dim3(256), 0, s, n_new_keys,
                       d_src.as<int64_t>(), d_off32.as<uint32_t>(), static_cast<const int64_t*>(c->b_loffs.p),
                       static_cast<const uint32_t*>(c->b_lpost.p), d_kind.as<uint8_t>(), uint32_t(cls),
                       static_cast<uint32_t*>(np.p), static_cast<int64_t*>(no.p), total);
    if (hipGetLastError() != hipSuccess) {
        drop();
        return ioc_fail(c, IOC_ERR_HIP, "k_upd_copy launch failed");
    }
    // ---- value sets: the cluster's segment becomes its new sorted set ----
    Tmp d_newset;
    if ((r = dev_alloc(c, d_newset, news.size() * 4)) != IOC_OK) {
        drop();
        return r;
    }
    if (!news.empty()) UCHK(c, hipMemcpyAsync(d_newset.p, news.data(), news.size() * 4, hipMemcpyHostToDevice, s));
    {
        const int64_t span = std::max<int64_t>(std::max<int64_t>(set_total, int64_t(news.size())), int64_t(c->L) + 1);
        hipLaunchKernelGGL(k_upd_sets, dim3(uint32_t((span + 255) / 256)), dim3(256), 0, s, c->L, cls,
                           static_cast<const int64_t*>(c->b_lset_off.p), static_cast<const uint32_t*>(c->b_lset_val.p), set_total,
                           d_newset.as<uint32_t>(), int64_t(news.size()), static_cast<int64_t*>(nso.p), static_cast<uint32_t*>(nsv.p));
        if (hipGetLastError() != hipSuccess) {
            drop();
            return ioc_fail(c, IOC_ERR_HIP, "k_upd_sets launch failed");
        }
    }
    UCHK(c, hipStreamSynchronize(s));
    // ---- swap the new state in ----
    for (DevBuf* b : {&c->b_lkeys, &c->b_loffs, &c->b_lpost, &c->b_lset_off, &c->b_lset_val})
        if (b->p) (void)hipFree(b->p);
    c->b_lkeys = nk;
    c->b_loffs = no;
    c->b_lpost = np;
    c->b_lset_off = nso;
    c->b_lset_val = nsv;
    if (c->b_lslot.cap < size_t(n_new_keys) * 4) {
        if (c->b_lslot.p) (void)hipFree(c->b_lslot.p);
        c->b_lslot.p = nullptr;
        c->b_lslot.cap = 0;
        if (hipMalloc(&c->b_lslot.p, size_t(n_new_keys) * 4 + 16) != hipSuccess)
            return ioc_fail(c, IOC_ERR_CAPACITY, "ioc_index_update: hipMalloc failed");
        c->b_lslot.cap = size_t(n_new_keys) * 4 + 16;
    }
    c->n_left_keys = n_new_keys;
    c->n_left_post = int64_t(total);
    const int64_t a = c->h_lset_off[size_t(cls)], b = c->h_lset_off[size_t(cls) + 1];
    const int64_t shift = int64_t(news.size()) - (b - a);
    for (size_t i = size_t(cls) + 1; i <= size_t(c->L); ++i) c->h_lset_off[i] += shift;
    c->built = c->scored = c->resolved = false;  // the combined index has to be rebuilt
    return IOC_OK;
}

int ioc_left_export(ioc_ctx* c, int64_t* n_keys, int64_t* n_postings, uint32_t* keys, int64_t* offs, uint32_t* postings)
{
    if (!c) return IOC_ERR_ARG;
    UCHK(c, hipSetDevice(c->device));
    if (n_keys) *n_keys = c->n_left_keys;
    if (n_postings) *n_postings = c->n_left_post;
    hipStream_t s = c->stream;
    if (keys && c->n_left_keys) UCHK(c, hipMemcpyAsync(keys, c->b_lkeys.p, size_t(c->n_left_keys) * 4, hipMemcpyDeviceToHost, s));
    if (offs) {
        if (c->n_left_keys)
            UCHK(c, hipMemcpyAsync(offs, c->b_loffs.p, size_t(c->n_left_keys + 1) * 8, hipMemcpyDeviceToHost, s));
        else
            offs[0] = 0;
    }
    if (postings && c->n_left_post) UCHK(c, hipMemcpyAsync(postings, c->b_lpost.p, size_t(c->n_left_post) * 4, hipMemcpyDeviceToHost, s));
    UCHK(c, hipStreamSynchronize(s));
    return IOC_OK;
}

}  // extern "C"
