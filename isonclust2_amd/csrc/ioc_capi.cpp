// ioc_capi.cpp — context management and the device-facing half of the C ABI
// (include/isonclust2_hip.h).  Host code only; kernels live in ioc_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "ioc_internal.h"
#include "ioc_kernels.h"

int ioc_fail(ioc_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

static uint32_t env_u32(const char* name, uint32_t dflt);

void ioc_poison(void* p, size_t bytes)
{
    static const int v = getenv("IOC_POISON") ? int(strtol(getenv("IOC_POISON"), nullptr, 0)) : -1;
    if (v >= 0 && p && bytes) (void)hipMemset(p, v & 0xFF, bytes);
}

int ioc_wait_uploads(ioc_ctx* c, int stage)
{
    if (!c) return IOC_ERR_ARG;
    while (c->up_stage.load(std::memory_order_acquire) < stage) std::this_thread::yield();
    if (c->up_failed.load(std::memory_order_acquire)) {
        // a failed copy is reported at the FIRST wait behind it (stage 1 included: the scoring must not run over a tail of
        // the value array that never arrived); the thread ends at once after a failure
        while (c->up_stage.load(std::memory_order_acquire) < 2) std::this_thread::yield();
        stage = 2;
    }
    if (stage >= 2 && c->up_thread.joinable()) c->up_thread.join();
    if (c->up_stage.load(std::memory_order_acquire) >= 2 && !c->up_thread.joinable() && !c->up_err.empty()) {
        const std::string m = c->up_err;
        c->up_err.clear();
        c->up_failed.store(false, std::memory_order_release);
        return ioc_fail(c, IOC_ERR_HIP, "upload of the query arrays: " + m);
    }
    return IOC_OK;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));   \
    } while (0)

static int dev_reserve(ioc_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return IOC_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return ioc_fail(c, IOC_ERR_CAPACITY,
                        "hipMalloc(" + std::to_string(want) + " B) failed: " + hipGetErrorString(e));
    }
    b.cap = want;
    ioc_poison(b.p, want);
    return IOC_OK;
}
#define RESERVE(c, b, bytes)                     \
    do {                                         \
        int r__ = dev_reserve((c), (b), (bytes)); \
        if (r__ != IOC_OK) return r__;           \
    } while (0)

static void dev_free(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

template <class T>
static T* P(DevBuf& b)
{
    return static_cast<T*>(b.p);
}

extern "C" {

int ioc_ctx_create(int device, ioc_ctx** out)
{
    if (!out) return IOC_ERR_ARG;
    *out = nullptr;
    // (IOC_TRACE: where a context's creation spends its time — the runtime's own start-up is most of it)
    const bool trace = getenv("IOC_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[ioc] ctx_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IOC_ERR_NO_DEVICE;
    lap("hipGetDeviceCount (hipInit)");
    if (device < 0 || device >= ndev) return IOC_ERR_ARG;
    ioc_ctx* c = new (std::nothrow) ioc_ctx;
    if (!c) return IOC_ERR_CAPACITY;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return IOC_ERR_HIP;
    }
    lap("hipSetDevice");
    if (hipStreamCreate(&c->own_stream) != hipSuccess) {
        delete c;
        return IOC_ERR_HIP;
    }
    lap("hipStreamCreate");
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), 256, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return IOC_ERR_HIP;
    }
    lap("hipHostMalloc");
    c->stream = c->own_stream;
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete c;
            return IOC_ERR_HIP;
        }
    lap("hipEventCreate x n");
    // k_score_part has two builds.  The DEFAULT tests every posting against the workgroup's target window (defined behaviour).
    // The other one has no window test and leans on gfx950 dropping LDS atomics beyond the workgroup's allocation; it bought
    // 1.6 % of the kernel's time on config 2 (the kernel is LDS-bound, not VALU-bound), so it is opt-in: IOC_SCORE_OOB=1 asks
    // for it, and even then only a passed probe on THIS device (k_lds_oob_probe) selects it.
    const char* force = getenv("IOC_SCORE_OOB");
    if (force && atoi(force) != 0) {
        uint32_t* d_res = nullptr;
        uint32_t h_res[2] = {~0u, 0u};
        if (hipMalloc(&d_res, 8) != hipSuccess || iock_lds_oob_probe(c->stream, d_res, h_res) != hipSuccess) {
            if (d_res) (void)hipFree(d_res);
            delete c;
            return IOC_ERR_HIP;
        }
        (void)hipFree(d_res);
        c->score_oob_probe = int(h_res[0]);
        c->score_oob = h_res[0] == 0;
        if (h_res[0] != 0 && getenv("IOC_TRACE"))
            fprintf(stderr, "[ioc] LDS out-of-bounds probe failed (%u): scoring keeps its window test\n", h_res[0]);
    }
    *out = c;
    return IOC_OK;
}

void ioc_ctx_destroy(ioc_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)ioc_wait_uploads(c, 2);
    (void)hipStreamSynchronize(c->stream);
    (void)ioc_dist_shutdown(c);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    for (auto& e : c->ev_side)
        if (e) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->b_off_fwd, &c->b_off_rev, &c->b_min, &c->b_pos, &c->b_hpc_len, &c->b_err_cell,
                      &c->b_min_total, &c->b_doff, &c->b_left_err, &c->b_lkeys, &c->b_loffs, &c->b_lpost,
                      &c->b_lslot, &c->b_lset_off, &c->b_lset_val, &c->b_keys, &c->b_cnt, &c->b_off, &c->b_fill,
                      &c->b_rows, &c->b_post, &c->b_dvals, &c->b_dcount, &c->b_dslot, &c->b_scan,
                      &c->b_cand_key, &c->b_cand_size, &c->b_cand_mapped, &c->b_cand_count, &c->b_valid0,
                      &c->b_valid1, &c->b_dec_target, &c->b_dec_strand, &c->b_flags, &c->b_forced_t,
                      &c->b_forced_s, &c->b_misc, &c->b_glim, &c->b_queue, &c->b_cut, &c->b_qinfo, &c->b_exp_cid, &c->b_exp_cnt, &c->b_exp_off, &c->b_exp_out, &c->b_exp_work, &c->b_dlong, &c->b_part, &c->b_shard_stage, &c->b_gap_bound, &c->b_keep_q, &c->b_bsort, &c->b_diag, &c->b_top_all, &c->b_pmins, &c->b_pbnd, &c->a_pool, &c->a_pairs, &c->a_order, &c->a_out, &c->a_bnd, &c->a_lrow, &c->a_ck, &c->a_cko, &c->a_ends, &c->a_ends2, &c->a_xflags, &c->a_prof, &c->b_aln_t, &c->b_aln_s, &c->b_tie_count, &c->b_tie_keys, &c->b_qhist, &c->b_qfirst, &c->b_qout, &c->b_qlist, &c->x_min, &c->x_pos, &c->x_off_fwd, &c->x_off_rev,
                      &c->x_hpc_len, &c->x_hseq, &c->x_hqual, &c->b_dist_min, &c->b_dist_pos};
    for (auto b : bufs) dev_free(*b);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_pin_big) (void)hipHostFree(c->h_pin_big);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int ioc_ctx_prewarm(ioc_ctx* c, int32_t alignment_mode)
{
    if (!c) return IOC_ERR_ARG;
    // A code object is loaded when one of its kernels is first launched: 2 - 25 ms each for the six files of the clustering path,
    // in the middle of a one-shot process's critical path.  Here they are loaded by threads of their own, beside the caller's
    // uploads (IOC_TRACE prints what each took).  Nothing depends on it: a launch that comes first loads its file itself.
    const int dev = c->device;
    const bool trace = getenv("IOC_TRACE") != nullptr;
    auto go = [dev, trace](const char* what, hipError_t (*fn)()) {
        std::thread([=] {
            const auto t0 = std::chrono::steady_clock::now();
            if (hipSetDevice(dev) == hipSuccess) (void)fn();
            if (trace)
                fprintf(stderr, "[ioc] prewarm: %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }).detach();
    };
    go("kernels", iock_warm_kernels);
    go("build_sort", iock_warm_build_sort);
    go("score", iock_warm_score);
    go("resolve", iock_warm_resolve);
    go("sort", iock_warm_sort);
    if (alignment_mode) go("align", iock_warm_align);
    return IOC_OK;
}

int ioc_ctx_trim(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->side_stream) HIPCHK(c, hipStreamSynchronize(c->side_stream));
    // the aligner's arenas: checkpoints (8 GB for config 3's batch), tables, profiles, traceback scratch
    DevBuf* bufs[] = {&c->a_ck, &c->a_cko, &c->a_prof, &c->a_bnd, &c->a_lrow, &c->a_xflags, &c->a_ends, &c->a_ends2};
    for (auto b : bufs) dev_free(*b);
    return IOC_OK;
}

const char* ioc_last_error(const ioc_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ioc_set_stream(ioc_ctx* c, void* s)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
    return IOC_OK;
}

int64_t ioc_queries_generation(const ioc_ctx* c) { return c ? int64_t(c->query_gen) : -1; }

int ioc_synchronize(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_set_params(ioc_ctx* c, const ioc_params* p, const int32_t gap_limit[225])
{
    if (!c || !p || !gap_limit) return IOC_ERR_ARG;
    if (p->k < 1 || p->k > 32 || p->w < p->k) return ioc_fail(c, IOC_ERR_ARG, "bad k/w");
    HIPCHK(c, hipSetDevice(c->device));
    c->params = *p;
    memcpy(c->h_glim, gap_limit, sizeof(c->h_glim));
    c->gap_bound_gen = ~0ull;  // (the bounds of totalMapped follow the gap limits)
    for (int i = 0; i < 225; ++i)
        if (gap_limit[i] < -1) return ioc_fail(c, IOC_ERR_ARG, "gap_limit < -1");
    // Candidates with Size < keep can never be walked: top >= MinShared is required
    // (cluster.cpp:376-379) and then cut = int(top * MinFraction) >= int(MinShared * MinFraction).
    int keep = 1;
    if (p->min_fraction >= 0.0 && p->min_fraction <= 1.0 && p->min_shared > 0) {
        keep = int(double(p->min_shared) * p->min_fraction);
        if (keep > p->min_shared) keep = p->min_shared;
        if (keep < 1) keep = 1;
    }
    c->keep = keep;
    RESERVE(c, c->b_glim, 225 * 4);
    HIPCHK(c, hipMemcpyAsync(c->b_glim.p, c->h_glim, 225 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_params = true;
    c->scored = c->resolved = false;
    return IOC_OK;
}

static int queries_common(ioc_ctx* c, int32_t n, const int64_t* off_fwd, const int64_t* off_rev, int64_t total)
{
    if (n < 0 || total < 0 || (n > 0 && (!off_fwd || !off_rev))) return ioc_fail(c, IOC_ERR_ARG, "bad query shape");
    c->chunked_call = false;  // (new queries: whatever a chunked ioc_cluster_merge left is gone; it sets the flag again when it returns)
    c->h_off_fwd.assign(off_fwd, off_fwd + n + 1);
    c->h_off_rev.assign(off_rev, off_rev + n + 1);
    c->h_doff.assign(size_t(n) + 1, 0);
    c->max_fwd = c->max_rev = 0;
    for (int i = 0; i < n; ++i) {
        int64_t a = off_fwd[i + 1] - off_fwd[i], b = off_rev[i + 1] - off_rev[i];
        if (a < 0 || b < 0 || off_fwd[i] < 0 || off_rev[i] < 0 || off_fwd[i + 1] > total || off_rev[i + 1] > total)
            return ioc_fail(c, IOC_ERR_ARG, "minimizer offsets out of range at query " + std::to_string(i));
        if (a > INT32_MAX / 4 || b > INT32_MAX / 4) return ioc_fail(c, IOC_ERR_CAPACITY, "query too long");
        c->max_fwd = std::max<int32_t>(c->max_fwd, int32_t(a));
        c->max_rev = std::max<int32_t>(c->max_rev, int32_t(b));
        c->h_doff[size_t(i) + 1] = c->h_doff[size_t(i)] + a;
    }
    c->n = n;
    c->total = total;
    ++c->query_gen;
    c->built = c->scored = c->resolved = false;
    c->h_forced_t.assign(size_t(n), INT32_MIN);
    c->h_forced_s.assign(size_t(n), 0);
    c->forced_dirty = true;
    c->forced_host_clear = true;
    c->forced_dev_clear = false;
    c->aln_verdicts = false;
    c->have_res_seq = false;
    RESERVE(c, c->b_doff, size_t(n + 1) * 8);
    HIPCHK(c, hipMemcpyAsync(c->b_doff.p, c->h_doff.data(), size_t(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    return IOC_OK;
}

int ioc_queries_upload(ioc_ctx* c, int32_t n, const int64_t* off_fwd, const int64_t* off_rev,
                       const uint32_t* min_val, const uint32_t* min_pos, int64_t total, const uint32_t* hpc_len,
                       const uint8_t* err_cell, const uint32_t* min_total)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0 && (!hpc_len || !err_cell || !min_total || (total > 0 && (!min_val || !min_pos))))
        return ioc_fail(c, IOC_ERR_ARG, "null query array");
    for (int i = 0; i < n; ++i)
        if (err_cell[i] < 1 || err_cell[i] > 15) return ioc_fail(c, IOC_ERR_ARG, "err_cell outside 1..15");
    int r = ioc_wait_uploads(c, 2);
    if (r != IOC_OK) return r;
    r = queries_common(c, n, off_fwd, off_rev, total);
    if (r != IOC_OK) return r;
    RESERVE(c, c->b_off_fwd, size_t(n + 1) * 8);
    RESERVE(c, c->b_off_rev, size_t(n + 1) * 8);
    RESERVE(c, c->b_min, size_t(total) * 4);
    RESERVE(c, c->b_pos, size_t(total) * 4);
    RESERVE(c, c->b_hpc_len, size_t(n) * 4);
    RESERVE(c, c->b_err_cell, size_t(n));
    RESERVE(c, c->b_min_total, size_t(n) * 4);
    hipStream_t s = c->stream;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_off_fwd.p, off_fwd, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_off_rev.p, off_rev, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_hpc_len.p, hpc_len, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_err_cell.p, err_cell, size_t(n), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_min_total.p, min_total, size_t(n) * 4, hipMemcpyHostToDevice, s));
    }
    // Inside ioc_cluster_merge (defer_uploads: the caller's arrays outlive the call) a large batch sends only what the
    // index build reads — the forward lists' values — here; the reverse lists' values (the scoring waits for them) and the
    // positions (the resolve does) follow on a copy stream from a thread of their own, under the first kernels.
    int64_t head = total;  // values uploaded here: [0, head)
    if (c->defer_uploads && total >= (int64_t(1) << 21) && env_u32("IOC_UPLOAD_OVERLAP", 1) == 1) {
        if (n > 0 && off_fwd[0] == 0 && off_rev[0] >= off_fwd[n]) head = off_fwd[n];  // the usual layout: [all forward][all reverse]
        if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    }
    if (head > 0) HIPCHK(c, hipMemcpyAsync(c->b_min.p, min_val, size_t(head) * 4, hipMemcpyHostToDevice, s));
    if (c->copy_stream && c->defer_uploads && total >= (int64_t(1) << 21) && env_u32("IOC_UPLOAD_OVERLAP", 1) == 1) {
        c->up_err.clear();
        c->up_failed.store(false, std::memory_order_release);
        c->up_stage.store(0, std::memory_order_release);
        uint32_t* d_min = P<uint32_t>(c->b_min);
        uint32_t* d_pos = P<uint32_t>(c->b_pos);
        const int dev = c->device;
        hipStream_t cs = c->copy_stream;
        c->up_thread = std::thread([c, dev, cs, d_min, d_pos, min_val, min_pos, head, total] {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess && total > head)
                e = hipMemcpyAsync(d_min + head, min_val + head, size_t(total - head) * 4, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipStreamSynchronize(cs);
            if (e != hipSuccess) {
                c->up_err = hipGetErrorString(e);
                c->up_failed.store(true, std::memory_order_release);
            }
            c->up_stage.store(1, std::memory_order_release);
            if (e == hipSuccess) e = hipMemcpyAsync(d_pos, min_pos, size_t(total) * 4, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipStreamSynchronize(cs);
            if (e != hipSuccess && c->up_err.empty()) {
                c->up_err = hipGetErrorString(e);
                c->up_failed.store(true, std::memory_order_release);
            }
            c->up_stage.store(2, std::memory_order_release);
        });
    } else if (total > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_pos.p, min_pos, size_t(total) * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    c->d_off_fwd = P<int64_t>(c->b_off_fwd);
    c->d_off_rev = P<int64_t>(c->b_off_rev);
    c->d_min = P<uint32_t>(c->b_min);
    c->d_pos = P<uint32_t>(c->b_pos);
    c->d_hpc_len = P<uint32_t>(c->b_hpc_len);
    c->d_err_cell = P<uint8_t>(c->b_err_cell);
    c->d_min_total = P<uint32_t>(c->b_min_total);
    c->borrowed = false;
    return IOC_OK;
}

// ioc_queries_upload with the two minimizer arrays already in HBM (borrowed, used in place)
int ioc_queries_upload_devmins(ioc_ctx* c, int32_t n, const int64_t* off_fwd, const int64_t* off_rev, const uint32_t* d_min_val,
                               const uint32_t* d_min_pos, int64_t total, const uint32_t* hpc_len, const uint8_t* err_cell,
                               const uint32_t* min_total)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0 && (!hpc_len || !err_cell || !min_total || (total > 0 && (!d_min_val || !d_min_pos))))
        return ioc_fail(c, IOC_ERR_ARG, "null query array");
    for (int i = 0; i < n; ++i)
        if (err_cell[i] < 1 || err_cell[i] > 15) return ioc_fail(c, IOC_ERR_ARG, "err_cell outside 1..15");
    int r = ioc_wait_uploads(c, 2);
    if (r != IOC_OK) return r;
    r = queries_common(c, n, off_fwd, off_rev, total);
    if (r != IOC_OK) return r;
    RESERVE(c, c->b_off_fwd, size_t(n + 1) * 8);
    RESERVE(c, c->b_off_rev, size_t(n + 1) * 8);
    RESERVE(c, c->b_hpc_len, size_t(n) * 4);
    RESERVE(c, c->b_err_cell, size_t(n));
    RESERVE(c, c->b_min_total, size_t(n) * 4);
    hipStream_t s = c->stream;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_off_fwd.p, off_fwd, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_off_rev.p, off_rev, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_hpc_len.p, hpc_len, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_err_cell.p, err_cell, size_t(n), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_min_total.p, min_total, size_t(n) * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    c->d_off_fwd = P<int64_t>(c->b_off_fwd);
    c->d_off_rev = P<int64_t>(c->b_off_rev);
    c->d_min = d_min_val;
    c->d_pos = d_min_pos;
    c->d_hpc_len = P<uint32_t>(c->b_hpc_len);
    c->d_err_cell = P<uint8_t>(c->b_err_cell);
    c->d_min_total = P<uint32_t>(c->b_min_total);
    c->borrowed = true;
    return IOC_OK;
}

int64_t ioc_gather_records_device(ioc_ctx* c, int32_t n_idx, const int32_t* entries, uint32_t* d_out_min, uint32_t* d_out_pos,
                                  int64_t cap, int64_t* off_fwd, int64_t* off_rev)
{
    if (!c || n_idx < 0 || (n_idx > 0 && (!entries || !off_fwd || !off_rev))) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->h_off_fwd.size() != size_t(c->n) + 1 || !c->d_min || !c->d_pos) return ioc_fail(c, IOC_ERR_STATE, "no queries on the device");
    if (c->chunked_call) return ioc_fail(c, IOC_ERR_STATE, "the last clustering call ran its batch in chunks: only the last chunk's entries are on the device");
    {
        const int rw = ioc_wait_uploads(c, 2);
        if (rw != IOC_OK) return rw;
    }
    std::vector<int64_t> src(size_t(2) * n_idx + 1), dst(size_t(2) * n_idx + 1);
    std::vector<uint32_t> len(size_t(2) * n_idx + 1);
    int64_t tot = 0;
    for (int side = 0; side < 2; ++side) {
        const std::vector<int64_t>& ho = side == 0 ? c->h_off_fwd : c->h_off_rev;
        int64_t* oo = side == 0 ? off_fwd : off_rev;
        for (int i = 0; i < n_idx; ++i) {
            const int e = entries[i];
            if (e < 0 || e >= c->n) return ioc_fail(c, IOC_ERR_ARG, "ioc_gather_records_device: entry out of range");
            const size_t x = size_t(side) * n_idx + size_t(i);
            src[x] = ho[size_t(e)];
            dst[x] = tot;
            len[x] = uint32_t(ho[size_t(e) + 1] - ho[size_t(e)]);
            oo[i] = tot;
            tot += int64_t(len[x]);
        }
        oo[n_idx] = tot;
    }
    if (tot > cap) return ioc_fail(c, IOC_ERR_CAPACITY, "ioc_gather_records_device: output buffers too small: need " + std::to_string(tot));
    if (tot == 0) return 0;
    if (!d_out_min || !d_out_pos) return ioc_fail(c, IOC_ERR_ARG, "null device buffer");
    const size_t nl = size_t(2) * n_idx;
    RESERVE(c, c->b_misc, nl * 20 + 256);
    int64_t* d_src = reinterpret_cast<int64_t*>(P<uint8_t>(c->b_misc) + 256);
    int64_t* d_dst = d_src + nl;
    uint32_t* d_len = reinterpret_cast<uint32_t*>(d_dst + nl);
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(d_src, src.data(), nl * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_dst, dst.data(), nl * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_len, len.data(), nl * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, iock_gather_lists(s, uint32_t(nl), d_src, d_dst, d_len, c->d_min, c->d_pos, d_out_min, d_out_pos));
    HIPCHK(c, hipStreamSynchronize(s));
    return tot;
}

int ioc_queries_bind_device(ioc_ctx* c, int32_t n, const int64_t* d_off_fwd, const int64_t* d_off_rev,
                            const uint32_t* d_min_val, const uint32_t* d_min_pos, int64_t total,
                            const uint32_t* d_hpc_len, const uint8_t* d_err_cell, const uint32_t* d_min_total,
                            const int64_t* h_off_fwd, const int64_t* h_off_rev)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > 0 && (!d_off_fwd || !d_off_rev || !d_hpc_len || !d_err_cell || !d_min_total ||
                  (total > 0 && (!d_min_val || !d_min_pos))))
        return ioc_fail(c, IOC_ERR_ARG, "null device array");
    int r = queries_common(c, n, h_off_fwd, h_off_rev, total);
    if (r != IOC_OK) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->d_off_fwd = d_off_fwd;
    c->d_off_rev = d_off_rev;
    c->d_min = d_min_val;
    c->d_pos = d_min_pos;
    c->d_hpc_len = d_hpc_len;
    c->d_err_cell = d_err_cell;
    c->d_min_total = d_min_total;
    c->borrowed = true;
    return IOC_OK;
}

int ioc_left_load(ioc_ctx* c, int32_t L, const uint8_t* cls_err_cell, int64_t n_keys, const uint32_t* keys,
                  const int64_t* offs, const uint32_t* postings)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (L < 0 || n_keys < 0) return ioc_fail(c, IOC_ERR_ARG, "negative size");
    c->built = c->scored = c->resolved = false;
    if (L == 0) n_keys = 0;  // a MinDB without clusters has nothing to match (cluster.cpp:92-94)
    if (n_keys > 0 && (!keys || !offs || !postings || !cls_err_cell)) return ioc_fail(c, IOC_ERR_ARG, "null left array");
    int64_t np = n_keys > 0 ? offs[n_keys] : 0;
    // validate + transpose: cluster -> sorted distinct values (the membership sets of getMappedRatio)
    std::vector<int64_t> soff(size_t(L) + 1, 0);
    for (int64_t i = 0; i < n_keys; ++i) {
        if (offs[i + 1] < offs[i]) return ioc_fail(c, IOC_ERR_ARG, "left offsets not monotone");
        if (i > 0 && keys[i] <= keys[i - 1]) return ioc_fail(c, IOC_ERR_ARG, "left keys must be strictly ascending");
        for (int64_t p = offs[i]; p < offs[i + 1]; ++p) {
            if (postings[p] >= uint32_t(L)) return ioc_fail(c, IOC_ERR_ARG, "left posting >= n_clusters");
            if (p > offs[i] && postings[p] <= postings[p - 1])
                return ioc_fail(c, IOC_ERR_ARG, "left posting lists must be strictly ascending");
            soff[postings[p] + 1]++;
        }
    }
    for (int i = 0; i < L; ++i) soff[size_t(i) + 1] += soff[size_t(i)];
    std::vector<uint32_t> sval(size_t(np > 0 ? np : 1));
    {
        std::vector<int64_t> cur(soff.begin(), soff.end() - 1);
        for (int64_t i = 0; i < n_keys; ++i)  // keys ascending -> each cluster's set comes out sorted
            for (int64_t p = offs[i]; p < offs[i + 1]; ++p) sval[size_t(cur[postings[p]]++)] = keys[i];
    }
    c->L = L;
    c->n_left_keys = n_keys;
    c->n_left_post = np;
    c->h_lset_off = soff;
    RESERVE(c, c->b_left_err, size_t(L));
    RESERVE(c, c->b_lkeys, size_t(n_keys) * 4);
    RESERVE(c, c->b_loffs, size_t(n_keys + 1) * 8);
    RESERVE(c, c->b_lpost, size_t(np) * 4);
    RESERVE(c, c->b_lslot, size_t(n_keys) * 4);
    RESERVE(c, c->b_lset_off, size_t(L + 1) * 8);
    RESERVE(c, c->b_lset_val, size_t(np) * 4);
    hipStream_t s = c->stream;
    if (L > 0) {
        for (int i = 0; i < L; ++i)
            if (cls_err_cell[i] < 1 || cls_err_cell[i] > 15) return ioc_fail(c, IOC_ERR_ARG, "left err_cell outside 1..15");
        HIPCHK(c, hipMemcpyAsync(c->b_left_err.p, cls_err_cell, size_t(L), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_lset_off.p, soff.data(), size_t(L + 1) * 8, hipMemcpyHostToDevice, s));
    }
    if (n_keys > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_lkeys.p, keys, size_t(n_keys) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_loffs.p, offs, size_t(n_keys + 1) * 8, hipMemcpyHostToDevice, s));
    }
    if (np > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_lpost.p, postings, size_t(np) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_lset_val.p, sval.data(), size_t(np) * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return IOC_OK;
}

static uint32_t env_u32(const char* name, uint32_t dflt)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    long x = strtol(v, nullptr, 10);
    return x >= 0 ? uint32_t(x) : dflt;
}

// k_gap_bounds for the current queries, if the table is not there yet: a function of the queries, the gap limits and
// the thresholds, kept across ioc_score calls on the same queries.  (Run on a stream of its own beside the index build it won
// 20 us of its 125: both fill the chip.)
static int gap_bounds_launch(ioc_ctx* c)
{
    const int n = c->n;
    const bool aln_mode_s = c->params.mode == IOC_MODE_SAHLIN || c->params.mode == IOC_MODE_FURIOUS;
    if (n <= 0 || env_u32("IOC_RESOLVE_BOUND", 1) != 1) return IOC_OK;
    const bool cut_lists = !aln_mode_s && env_u32("IOC_SCORE_KEEPQ", 1) == 1;
    if (c->gap_bound_gen == c->query_gen && c->gap_bound_cut == cut_lists) return IOC_OK;
    {
        const int rw = ioc_wait_uploads(c, 2);  // the positions
        if (rw != IOC_OK) return rw;
    }
    RESERVE(c, c->b_gap_bound, size_t(n) * 2 * 15 * sizeof(uint2));
    RESERVE(c, c->b_keep_q, size_t(n) * 4);
    HIPCHK(c, iock_gap_bounds(c->stream, n, c->d_off_fwd, c->d_off_rev, c->d_pos, c->d_hpc_len, c->d_err_cell, P<int32_t>(c->b_glim),
                              P<uint2>(c->b_gap_bound), c->d_min_total, uint32_t(c->keep), cut_lists ? P<uint32_t>(c->b_keep_q) : nullptr));
    c->gap_bound_gen = c->query_gen;
    c->gap_bound_cut = cut_lists;
    return IOC_OK;
}

int ioc_index_build(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_params) return ioc_fail(c, IOC_ERR_STATE, "ioc_set_params first");
    const int n = c->n;
    const int64_t nfwd_total = c->h_doff.empty() ? 0 : c->h_doff[size_t(n)];
    const int64_t ub_entries = nfwd_total + c->n_left_post;
    if (ub_entries >= (int64_t(1) << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 2^31 index postings");
    if (uint64_t(c->L) + uint64_t(n) >= (1ull << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "too many targets");
    if (uint32_t(n) > 131072u) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 131072 queries per device pass (ioc_cluster_merge runs a larger batch in chunks)");
    uint32_t pmax = 1;
    while (pmax < uint32_t(c->max_fwd)) pmax <<= 1;
    // (a query of more than IOC_DISTINCT_LDS_MAX forward minimizers takes the long way round below; IOC_DISTINCT_BITONIC=1, the
    // round-1 network kept for comparison, sorts up to 32768 in LDS)
    if (pmax > 32768 && env_u32("IOC_DISTINCT_BITONIC", 0) == 1) return ioc_fail(c, IOC_ERR_CAPACITY, "IOC_DISTINCT_BITONIC=1: a query has more than 32768 forward minimizers");
    hipStream_t s = c->stream;
    c->tm.ms_align_fwd = c->tm.ms_align_trace = 0.f;
    c->tm.n_align_pairs = c->tm.n_align_cells = c->tm.n_align_refused = c->tm.n_align_cells_computed = 0;
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    // 16-bit postings when every target id fits (the padding value 0xFFFF must stay above every id)
    c->post16 = (uint64_t(c->L) + uint64_t(n) <= 65535ull && env_u32("IOC_POST16", 1) == 1) ? 1 : 0;
    const uint32_t psize = c->post16 ? 2u : 4u;
    const uint32_t pmask = 16u / psize - 1u;  // lists are padded to whole 16-byte units

    RESERVE(c, c->b_dvals, size_t(nfwd_total) * 4);
    RESERVE(c, c->b_dslot, size_t(nfwd_total) * 4);
    RESERVE(c, c->b_fill, size_t(nfwd_total) * 4);  // position of each distinct value inside its posting list
    RESERVE(c, c->b_dcount, size_t(n) * 4);
    RESERVE(c, c->b_misc, 256);
    // ---- hash table sizing: distinct keys <= min(entries, 4^k); HPC sequences have no equal
    // neighbours, so at most 4*3^(k-1) distinct k-mers occur — used as the first guess only.
    const int k = c->params.k;
    double ub = double(nfwd_total + c->n_left_keys);
    if (k <= 15) ub = std::min(ub, std::pow(4.0, k));
    double guess = ub;
    if (k <= 20) guess = std::min(guess, 4.0 * std::pow(3.0, k - 1));
    auto pow2_at_least = [](double x) {
        uint32_t cap = 1024;
        while (double(cap) < x && cap < (1u << 30)) cap <<= 1;
        return cap;
    };
    uint32_t cap = pow2_at_least(2.0 * guess);
    const uint32_t cap_safe = pow2_at_least(2.0 * ub);
    // ---- the build without global atomics (ioc_build_sort.hip): a stable radix sort of the (value, target) pairs ----
    const int value_bits = (k >= 1 && k <= 16) ? 2 * k : 32;
    const bool sorted_build = value_bits < 32 && ub_entries > 0 && env_u32("IOC_BUILD_SORT", 1) == 1;
    const int64_t NP = ub_entries;  // (the unused tails of the queries' lists ride along as sentinels)
    IocBuildSort a{};
    if (sorted_build) {
        a.n = n;
        a.L = uint32_t(c->L);
        a.doff = P<int64_t>(c->b_doff);
        a.dcount = P<uint32_t>(c->b_dcount);
        a.dvals = P<uint32_t>(c->b_dvals);
        a.n_left_keys = c->n_left_keys;
        a.n_left_post = c->n_left_post;
        a.lkeys = P<uint32_t>(c->b_lkeys);
        a.loffs = P<int64_t>(c->b_loffs);
        a.lpost = P<uint32_t>(c->b_lpost);
        a.P = NP;
        a.post16 = c->post16;
        a.value_bits = value_bits;
        a.pad_mask = pmask;
        a.temp_bytes = iock_build_sort_temp_bytes(NP, c->post16, value_bits);
        // one arena: [pk_in][pk_out][rid][roff][run_start][lens][scan scratch][ctl][pv_in][pv_out][temp]
        const size_t w = (size_t(NP) + 7) & ~size_t(3);  // (every array 16-byte aligned: the run numbering reads and writes vectors)
        // (the scan of phase 2 runs over the table's SLOTS, not over the pairs: up to cap_safe + 1 of them once the table has grown)
        const size_t scan_words = (std::max<size_t>(size_t(NP), size_t(cap_safe)) + 1) / 1024 + 8;
        const size_t words = 6 * w + scan_words + 4;
        const size_t pvb = (size_t(NP) * psize + 255) & ~size_t(255);
        RESERVE(c, c->b_bsort, words * 4 + 2 * pvb + a.temp_bytes + 1024);
        uint32_t* wp = P<uint32_t>(c->b_bsort);
        a.pk_in = wp;
        a.pk_out = wp + w;
        a.rid = wp + 2 * w;
        a.run_slot = wp + 3 * w;
        a.run_start = wp + 4 * w;
        a.scan_scratch = wp + 6 * w;
        a.scan_words = scan_words;
        a.ctl = a.scan_scratch + scan_words;
        uint8_t* bp = reinterpret_cast<uint8_t*>(wp) + ((words * 4 + 255) & ~size_t(255));
        a.pv_in = bp;
        a.pv_out = bp + pvb;
        a.temp = bp + 2 * pvb;
    }
    {   // distinct values per query; with the sorted build the same kernel writes the queries' (value, target) pairs
        int written = 0;
        const uint32_t sentinel = value_bits < 32 ? 1u << value_bits : 0u;
        uint32_t* qk = sorted_build ? a.pk_in + c->n_left_post : nullptr;
        void* qv = sorted_build ? static_cast<void*>(static_cast<uint8_t*>(a.pv_in) + size_t(c->n_left_post) * psize) : nullptr;
        HIPCHK(c, iock_distinct(s, n, c->d_off_fwd, c->d_min, P<int64_t>(c->b_doff), P<uint32_t>(c->b_dvals), P<uint32_t>(c->b_dcount), pmax,
                                value_bits, qk, qv, c->post16, uint32_t(c->L), sentinel, &written));
        a.pairs_done = written;
        if (pmax > IOC_DISTINCT_LDS_MAX && env_u32("IOC_DISTINCT_BITONIC", 0) != 1) {
            // the queries that kernel left out: sorted in global memory (ioc_sort_long.hip)
            std::vector<int32_t> qid;
            std::vector<unsigned long long> seg(1, 0ull);
            for (int j = 0; j < n; ++j) {
                const int64_t m = c->h_off_fwd[size_t(j) + 1] - c->h_off_fwd[size_t(j)];
                if (m > int64_t(IOC_DISTINCT_LDS_MAX)) {
                    qid.push_back(j);
                    seg.push_back(seg.back() + (unsigned long long)m);
                }
            }
            const size_t total = size_t(seg.back()), nl = qid.size();
            if (total >= (size_t(1) << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 2^31 forward minimizers in the batch's long queries");
            const size_t tmpb = iock_distinct_long_temp(total, uint32_t(nl), value_bits);
            const size_t o_seg = (nl * 4 + 255) & ~size_t(255), o_work = o_seg + (((nl + 1) * 8 + 255) & ~size_t(255)), o_tmp = o_work + ((2 * total * 4 + 255) & ~size_t(255));
            RESERVE(c, c->b_dlong, o_tmp + tmpb);
            uint8_t* w = P<uint8_t>(c->b_dlong);
            HIPCHK(c, hipMemcpyAsync(w, qid.data(), nl * 4, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(w + o_seg, seg.data(), (nl + 1) * 8, hipMemcpyHostToDevice, s));
            HIPCHK(c, iock_distinct_long(s, uint32_t(nl), total, reinterpret_cast<const int32_t*>(w), reinterpret_cast<const unsigned long long*>(w + o_seg),
                                         c->d_off_fwd, c->d_min, P<int64_t>(c->b_doff), P<uint32_t>(c->b_dvals), P<uint32_t>(c->b_dcount), value_bits,
                                         reinterpret_cast<uint32_t*>(w + o_work), w + o_tmp, tmpb, written ? qk : nullptr, written ? qv : nullptr, c->post16,
                                         uint32_t(c->L), sentinel));
            HIPCHK(c, hipStreamSynchronize(s));  // (qid / seg are this frame's)
        }
    }
    if (sorted_build) {
        // Everything whose size does not hang on the sort's outcome is queued BEFORE the one read-back: the table at the capacity
        // the k-mer space suggests (twice the distinct keys that can occur), the postings at their upper bound (every pair a list of
        // its own, padded).  The host only has to queue the two kernels of phase 2 afterwards.
        uint32_t nslots = cap + 1;
        // (every pair + the padding of as many lists as the table is sized for; more lists than that: the slow way round below)
        const size_t post_ub = size_t(NP) + size_t(pmask) * std::min<size_t>(size_t(NP), size_t(cap) / 2 + 1) + 64;
        // (the table, the counters, the per-key query info, the control words and the postings are cleared by ONE launch)
        auto table = [&](uint32_t slots, size_t post_bytes) -> int {
            RESERVE(c, c->b_keys, size_t(slots) * 4);
            RESERVE(c, c->b_cnt, size_t(slots + 1) * 4);
            RESERVE(c, c->b_off, size_t(slots + 1) * 4);
            RESERVE(c, c->b_rows, size_t(slots) * 16);
            RESERVE(c, c->b_qinfo, size_t(slots) * 8);
            if (post_bytes) RESERVE(c, c->b_post, post_bytes);
            void* ptrs[5] = {c->b_keys.p, c->b_cnt.p, c->b_qinfo.p, c->b_misc.p, c->b_post.p};
            const size_t bytes[5] = {size_t(slots) * 4, size_t(slots + 1) * 4, size_t(slots) * 8, 256, post_bytes};
            const uint32_t vals[5] = {0xFFFFFFFFu, 0u, 0u, 0u, 0xFFFFFFFFu};
            HIPCHK(c, iock_fill_multi(s, post_bytes ? 5 : 4, ptrs, bytes, vals));
            return IOC_OK;
        };
        {
            const int rt = table(nslots, (post_ub * psize + 256 + 3) & ~size_t(3));
            if (rt != IOC_OK) return rt;
        }
        HIPCHK(c, iock_build_sort_phase1(s, &a));
        HIPCHK(c, hipMemcpyAsync(c->h_pin + 8, a.ctl, 8, hipMemcpyDeviceToHost, s));  // real pairs, runs = distinct keys
        HIPCHK(c, hipStreamSynchronize(s));
        const uint32_t n_real = static_cast<volatile uint32_t*>(c->h_pin)[8], R = static_cast<volatile uint32_t*>(c->h_pin)[9];
        const uint64_t h_total = uint64_t(n_real) + uint64_t(pmask) * R;  // (an upper bound of the padded postings; the exact figure: ioc_get_timings)
        if (h_total >= (1ull << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 2^31 padded index postings");
        if (size_t(h_total) > post_ub) {  // (more lists than the k-mer space suggested)
            RESERVE(c, c->b_post, size_t(h_total) * psize + 256);
            HIPCHK(c, hipMemsetAsync(c->b_post.p, 0xFF, size_t(h_total) * psize + 256, s));
        }
        if (2.0 * double(R) > double(cap)) {  // (more distinct keys than the k-mer space suggested: a larger table)
            cap = pow2_at_least(2.0 * double(R));
            nslots = cap + 1;
            const int rt = table(nslots, 0);
            if (rt != IOC_OK) return rt;
        }
        uint32_t bits = 0;
        while ((1u << bits) < cap) bits++;
        const uint32_t shift = 32 - bits;
        c->cap = cap;
        c->n_post = -1;  // (b_off[nslots]: on its way into pinned memory behind phase 2, read when the timings are asked for)
        HIPCHK(c, iock_build_sort_phase2(s, &a, R, n_real, P<uint32_t>(c->b_keys), cap, shift, P<uint32_t>(c->b_cnt), P<uint32_t>(c->b_off), c->b_post.p,
                                         P<uint32_t>(c->b_qinfo), uint32_t(n > 0 ? n : 1), P<uint32_t>(c->b_misc)));
        HIPCHK(c, iock_pack_rows(s, nslots, P<uint32_t>(c->b_keys), P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt), P<uint32_t>(c->b_qinfo), c->b_rows.p));
        HIPCHK(c, hipMemcpyAsync(c->h_pin + 40, P<uint32_t>(c->b_off) + nslots, 4, hipMemcpyDeviceToHost, s));
    }
    for (; !sorted_build;) {
        const uint32_t nslots = cap + 1;
        uint32_t bits = 0;
        while ((1u << bits) < cap) bits++;
        const uint32_t shift = 32 - bits;
        RESERVE(c, c->b_keys, size_t(nslots) * 4);
        RESERVE(c, c->b_cnt, size_t(nslots + 1) * 4);
        RESERVE(c, c->b_off, size_t(nslots + 1) * 4);
        RESERVE(c, c->b_rows, size_t(nslots) * 16);
        RESERVE(c, c->b_qinfo, size_t(nslots) * 8);  // two words per slot: length + epoch cuts (ioc_kernels.hip, index_lookup)
        RESERVE(c, c->b_scan, (size_t(nslots) / 1024 + 4) * 4);
        HIPCHK(c, hipMemsetAsync(c->b_keys.p, 0xFF, size_t(nslots) * 4, s));
        HIPCHK(c, hipMemsetAsync(c->b_cnt.p, 0, size_t(nslots + 1) * 4, s));
        HIPCHK(c, hipMemsetAsync(c->b_misc.p, 0, 256, s));
        uint32_t* d_err = P<uint32_t>(c->b_misc);
        HIPCHK(c, iock_hash_insert_left(s, c->n_left_keys, P<uint32_t>(c->b_lkeys), P<int64_t>(c->b_loffs),
                                        P<uint32_t>(c->b_keys), cap, shift, P<uint32_t>(c->b_cnt),
                                        P<uint32_t>(c->b_lslot), d_err));
        HIPCHK(c, iock_hash_insert_queries(s, n, P<int64_t>(c->b_doff), P<uint32_t>(c->b_dvals),
                                           P<uint32_t>(c->b_dcount), P<uint32_t>(c->b_keys), cap, shift,
                                           P<uint32_t>(c->b_cnt), P<uint32_t>(c->b_dslot), P<uint32_t>(c->b_fill),
                                           d_err));
        // posting lists start 16-byte aligned and are padded to a multiple of 4 entries (0xFFFFFFFF)
        HIPCHK(c, iock_exclusive_scan(s, P<uint32_t>(c->b_cnt), nslots, P<uint32_t>(c->b_off), P<uint32_t>(c->b_scan), pmask));
        // (into pinned memory: two pageable 4-byte read-backs cost ~60 us of host time between them)
        HIPCHK(c, hipMemcpyAsync(c->h_pin + 8, d_err, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_pin + 9, P<uint32_t>(c->b_off) + nslots, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const uint32_t h_err = static_cast<volatile uint32_t*>(c->h_pin)[8], h_total = static_cast<volatile uint32_t*>(c->h_pin)[9];
        if (h_err != 0) {
            if (cap >= cap_safe) return ioc_fail(c, IOC_ERR_CAPACITY, "hash table overflow at safe capacity");
            cap = std::min<uint64_t>(uint64_t(cap) * 4, cap_safe);
            continue;
        }
        c->cap = cap;
        c->n_post = h_total;
        if (uint64_t(h_total) >= (1ull << 31)) return ioc_fail(c, IOC_ERR_CAPACITY, "more than 2^31 padded index postings");
        RESERVE(c, c->b_post, size_t(h_total) * psize + 256);
        HIPCHK(c, hipMemsetAsync(c->b_post.p, 0xFF, size_t(h_total) * psize + 256, s));
        HIPCHK(c, iock_fill_left(s, c->n_left_keys, P<int64_t>(c->b_loffs), P<uint32_t>(c->b_lpost),
                                 P<uint32_t>(c->b_lslot), P<uint32_t>(c->b_off), c->b_post.p, c->post16));
        HIPCHK(c, iock_fill_queries(s, n, uint32_t(c->L), P<int64_t>(c->b_doff), P<uint32_t>(c->b_dcount),
                                    P<uint32_t>(c->b_dslot), P<uint32_t>(c->b_fill), P<uint32_t>(c->b_off),
                                    c->b_post.p, c->post16));
        HIPCHK(c, iock_sort_lists(s, nslots, P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt), c->b_post.p,
                                  uint32_t(c->L), uint32_t(n > 0 ? n : 1), 2048, P<uint32_t>(c->b_qinfo), c->post16));
        HIPCHK(c, iock_pack_rows(s, nslots, P<uint32_t>(c->b_keys), P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt),
                                 P<uint32_t>(c->b_qinfo), c->b_rows.p));
        break;
    }
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    c->built = true;
    c->scored = c->resolved = false;
    c->tm.n_queries = n;
    c->tm.n_minimizers = 0;
    for (int i = 0; i < n; ++i)
        c->tm.n_minimizers += (c->h_off_fwd[size_t(i) + 1] - c->h_off_fwd[size_t(i)]) +
                              (c->h_off_rev[size_t(i) + 1] - c->h_off_rev[size_t(i)]);
    c->tm.n_index_postings = c->n_post;
    return IOC_OK;
}

static uint32_t hash_shift(uint32_t cap)
{
    uint32_t bits = 0;
    while ((1u << bits) < cap) bits++;
    return 32 - bits;
}

int ioc_score(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->built) return ioc_fail(c, IOC_ERR_STATE, "ioc_index_build first");
    {
        const int rw = ioc_wait_uploads(c, 1);  // the reverse lists' values may still be on their way (ioc_queries_upload)
        if (rw != IOC_OK) return rw;
    }
    const int n = c->n;
    const uint64_t L = uint64_t(c->L);
    const uint64_t capacity = 2ull * L * uint64_t(n) + uint64_t(n) * uint64_t(n > 0 ? n - 1 : 0);
    size_t freeB = 0, totalB = 0;
    HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
    const uint64_t need = capacity * 12ull;
    const uint64_t have = uint64_t(freeB) + c->b_cand_key.cap + c->b_cand_size.cap + c->b_cand_mapped.cap;
    if (need + (1ull << 28) > have)
        return ioc_fail(c, IOC_ERR_CAPACITY, "candidate tables need " + std::to_string(need >> 20) + " MiB of HBM");
    RESERVE(c, c->b_cand_key, size_t(capacity) * 4);
    RESERVE(c, c->b_cand_size, size_t(capacity) * 4);
    RESERVE(c, c->b_cand_mapped, size_t(capacity) * 4);
    RESERVE(c, c->b_cand_count, size_t(n) * 4);
    RESERVE(c, c->b_top_all, size_t(n) * 4);
    RESERVE(c, c->b_misc, 256);
    c->cand_capacity = int64_t(capacity);
    hipStream_t s = c->stream;
    const bool count_trav = getenv("IOC_COUNT_TRAVERSED") != nullptr;
    unsigned long long* d_trav = reinterpret_cast<unsigned long long*>(P<uint8_t>(c->b_misc) + 64);
    HIPCHK(c, hipMemsetAsync(c->b_misc.p, 0, 256, s));
    HIPCHK(c, hipMemsetAsync(c->b_cand_mapped.p, 0xFF, size_t(capacity) * 4, s));
    const uint32_t range = env_u32("IOC_SCORE_RANGE", 8192);
    // XCD-partitioned scoring keeps 8 partial histograms per query (single-pass case only)
    uint32_t* d_part = nullptr;
    iock_set_score_variant(int(env_u32("IOC_SCORE_VARIANT", 0)));
    iock_set_part32(int(env_u32("IOC_PART32", 0)));
    iock_set_score_oob(c->score_oob);
    const bool aln_mode_s = c->params.mode == IOC_MODE_SAHLIN || c->params.mode == IOC_MODE_FURIOUS;
    // Upper bounds of totalMapped (k_gap_bounds): the sweeps of ioc_resolve reject candidates by them, and in fast mode the
    // candidate lists are cut at the smallest Size that passes any bound of the query (below it nothing passes, and the top
    // Size only matters when a candidate at or above it exists).  The alignment modes keep every candidate: the tie set of
    // the fallback is made of candidates that fail the mapping.
    c->keep_q_on = false;
    c->h_keep_q.clear();
    if (n > 0 && env_u32("IOC_RESOLVE_BOUND", 1) == 1) {
        const int rg = gap_bounds_launch(c);
        if (rg != IOC_OK) return rg;
        c->keep_q_on = c->gap_bound_cut;
    } else {
        c->gap_bound_gen = ~0ull;
    }
    // (the launcher's per-thread settings are taken back on EVERY way out of this function, the error returns included)
    struct ScoreSettings {
        ~ScoreSettings()
        {
            iock_set_score_shard(1, 0);
            iock_set_score_keep(nullptr);
        }
    } score_settings_guard;
    iock_set_score_keep(c->keep_q_on ? P<uint32_t>(c->b_keep_q) : nullptr);
    c->scored_sharded = c->shard_world > 1 && c->shard_fn && !aln_mode_s;
    iock_set_score_shard(c->scored_sharded ? c->shard_world : 1, c->shard_rank);
    if (env_u32("IOC_SCORE_PARTS", 1) == 1 && L + uint64_t(n) <= range && capacity * 8 * 4 + (1ull << 28) < have - need) {
        RESERVE(c, c->b_part, size_t(capacity) * 8 * 4);
        RESERVE(c, c->b_pmins, size_t(c->total) * 4);
        RESERVE(c, c->b_pbnd, (size_t(n) * 2 * 9 + 1) * 4);
        d_part = P<uint32_t>(c->b_part);
    } else if (c->b_part.p) {
        HIPCHK(c, hipStreamSynchronize(s));
        dev_free(c->b_part);
    }
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    HIPCHK(c, iock_score(s, n, uint32_t(L), c->d_off_fwd, c->d_off_rev, c->d_min, c->b_rows.p, c->cap,
                         hash_shift(c->cap), c->b_post.p, range, uint32_t(c->keep),
                         P<uint32_t>(c->b_cand_key), P<uint32_t>(c->b_cand_size), P<uint32_t>(c->b_cand_count),
                         count_trav ? d_trav : nullptr, nullptr, nullptr, d_part, P<uint32_t>(c->b_top_all), c->post16,
                         P<uint32_t>(c->b_pmins), P<uint32_t>(c->b_pbnd)));
    iock_set_score_shard(1, 0);
    iock_set_score_keep(nullptr);
    c->have_guess = d_part != nullptr && !c->scored_sharded;  // (b_top_all holds the owned queries only)
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (count_trav) {
        unsigned long long t = 0;
        HIPCHK(c, hipMemcpyAsync(&t, d_trav, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        c->tm.postings_traversed = int64_t(t);
    }
    c->scored = true;
    c->resolved = false;
    return IOC_OK;
}

int ioc_force_decision(ioc_ctx* c, int32_t q, int32_t target, int32_t strand)
{
    if (!c || q < 0 || q >= c->n) return IOC_ERR_ARG;
    if (target >= 0 && strand != 1 && strand != -1) return ioc_fail(c, IOC_ERR_ARG, "strand must be +1/-1");
    if (target >= c->L + q) return ioc_fail(c, IOC_ERR_ARG, "forced target is not an earlier target");
    c->h_forced_t[size_t(q)] = target < 0 ? (target == -2 ? -2 : -1) : target;  // -2: excluded (gated) entry
    c->h_forced_s[size_t(q)] = int8_t(target < 0 ? 0 : strand);
    c->forced_dirty = true;
    c->forced_host_clear = false;
    c->warm_first = -1;
    return IOC_OK;
}

int ioc_clear_forced(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    c->warm_first = -1;
    if (c->forced_host_clear && (c->forced_dev_clear || c->forced_dirty)) return IOC_OK;  // (cleared already, here and — or soon — there)
    std::fill(c->h_forced_t.begin(), c->h_forced_t.end(), INT32_MIN);
    std::fill(c->h_forced_s.begin(), c->h_forced_s.end(), 0);
    c->forced_host_clear = true;
    c->forced_dirty = true;
    c->warm_first = -1;
    return IOC_OK;
}

// Verdicts of the alignment fallback (getBestClusterAln, cluster.cpp:461-515), one per query: used by
// ioc_resolve only for a query whose mapping walk finds nothing although top >= MinShared.
int ioc_set_aln_verdicts(ioc_ctx* c, const int32_t* target, const int8_t* strand)
{
    if (!c) return IOC_ERR_ARG;
    if (!target) {
        c->aln_verdicts = false;
        c->warm_first = -1;
        return IOC_OK;
    }
    if (!strand) return IOC_ERR_ARG;
    const size_t n = size_t(c->n);
    for (size_t i = 0; i < n; ++i) {
        if (target[i] != INT32_MIN && target[i] >= c->L + int32_t(i))
            return ioc_fail(c, IOC_ERR_ARG, "alignment verdict is not an earlier target");
        if (target[i] >= 0 && strand[i] != 1 && strand[i] != -1) return ioc_fail(c, IOC_ERR_ARG, "strand must be +1/-1");
    }
    if (c->resolved && c->aln_verdicts && c->warm_first >= 0 && c->h_aln_t.size() == n && c->h_aln_s.size() == n) {
        size_t fd = 0;
        while (fd < n && c->h_aln_t[fd] == target[fd] && c->h_aln_s[fd] == strand[fd]) ++fd;
        c->warm_first = std::min<int32_t>(c->warm_first, int32_t(fd));
    } else {
        c->warm_first = -1;
    }
    c->h_aln_t.assign(target, target + n);
    c->h_aln_s.assign(strand, strand + n);
    c->aln_verdicts = true;
    c->aln_dirty = true;
    return IOC_OK;
}

// Candidates tied at the top Size among the current clusters, per query, as of the last ioc_resolve
// (only collected while verdicts are set): count[n], keys[n * 4] = target << 1 | (strand == -1).
int ioc_get_ties(ioc_ctx* c, uint32_t* count, uint32_t* keys)
{
    if (!c || !count || !keys) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved || !c->aln_verdicts) return ioc_fail(c, IOC_ERR_STATE, "ioc_set_aln_verdicts + ioc_resolve first");
    const size_t n = size_t(c->n);
    if (n == 0) return IOC_OK;
    HIPCHK(c, hipMemcpyAsync(count, c->b_tie_count.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(keys, c->b_tie_keys.p, n * IOC_TIE_SLOTS * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_set_shard(ioc_ctx* c, int32_t world, int32_t rank, ioc_exchange_fn fn, void* user)
{
    if (!c) return IOC_ERR_ARG;
    if (world > 1 && fn) {
        if (rank < 0 || rank >= world) return ioc_fail(c, IOC_ERR_ARG, "ioc_set_shard: rank outside the world");
        c->shard_world = world;
        c->shard_rank = rank;
        c->shard_fn = fn;
        c->shard_user = user;
        c->shard_aln_pairs = 0;
    } else {
        c->shard_world = 1;
        c->shard_rank = 0;
        c->shard_fn = nullptr;
        c->shard_user = nullptr;
    }
    c->scored = false;  // (candidate tables of the other setting)
    c->resolved = false;
    return IOC_OK;
}

int32_t ioc_shard_exchanges(const ioc_ctx* c) { return c ? c->shard_exchanges : 0; }
int64_t ioc_shard_aligned_pairs(const ioc_ctx* c) { return c ? c->shard_aln_pairs : 0; }

}  // extern "C"

// one all-reduce of the sharded path through the caller's hook, on the context's stream
int ioc_shard_exchange(ioc_ctx* c, void* d_buf, int64_t count, int kind)
{
    if (!c->shard_fn) return ioc_fail(c, IOC_ERR_STATE, "ioc_set_shard: no exchange installed");
    c->shard_exchanges++;
    if (c->shard_fn(c->shard_user, d_buf, count, kind, (void*)c->stream) != 0)
        return ioc_fail(c, IOC_ERR_STATE, "ioc_set_shard: the exchange callback failed");
    return IOC_OK;
}

// a host array of words summed over the ranks (every rank fills the slots it owns and leaves zeros elsewhere)
int ioc_shard_sum_host(ioc_ctx* c, int32_t* words, int64_t count)
{
    if (count <= 0) return IOC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    RESERVE(c, c->b_shard_stage, size_t(count) * 4);
    HIPCHK(c, hipMemcpyAsync(c->b_shard_stage.p, words, size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
    if (int rc = ioc_shard_exchange(c, c->b_shard_stage.p, count, IOC_XCHG_SUM_I32)) return rc;
    HIPCHK(c, hipMemcpyAsync(words, c->b_shard_stage.p, size_t(count) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

extern "C" {

int ioc_resolve(ioc_ctx* c, int32_t* n_iter)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->scored) return ioc_fail(c, IOC_ERR_STATE, "ioc_score first");
    {
        const int rw = ioc_wait_uploads(c, 2);  // the positions (ioc_queries_upload)
        if (rw != IOC_OK) return rw;
    }
    const int n = c->n;
    hipStream_t s = c->stream;
    RESERVE(c, c->b_valid0, size_t(n));
    RESERVE(c, c->b_valid1, size_t(n));
    RESERVE(c, c->b_dec_target, size_t(n) * 4);
    RESERVE(c, c->b_dec_strand, size_t(n));
    RESERVE(c, c->b_flags, size_t(n));
    RESERVE(c, c->b_forced_t, size_t(n) * 4);
    RESERVE(c, c->b_forced_s, size_t(n));
    RESERVE(c, c->b_misc, 256);
    if (c->forced_dirty && n > 0) {
        HIPCHK(c, hipMemcpyAsync(c->b_forced_t.p, c->h_forced_t.data(), size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_forced_s.p, c->h_forced_s.data(), size_t(n), hipMemcpyHostToDevice, s));
        c->forced_dirty = false;
        c->forced_dev_clear = c->forced_host_clear;
    }
    if (c->aln_verdicts) {
        RESERVE(c, c->b_aln_t, size_t(n) * 4);
        RESERVE(c, c->b_aln_s, size_t(n));
        RESERVE(c, c->b_tie_count, size_t(n) * 4);
        RESERVE(c, c->b_tie_keys, size_t(n) * IOC_TIE_SLOTS * 4);
        if (c->aln_dirty && n > 0) {
            HIPCHK(c, hipMemcpyAsync(c->b_aln_t.p, c->h_aln_t.data(), size_t(n) * 4, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(c->b_aln_s.p, c->h_aln_s.data(), size_t(n), hipMemcpyHostToDevice, s));
            c->aln_dirty = false;
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[4], s));
    // Warm start: the previous call's fixed point is still in the buffers and only alignment verdicts changed
    // since, the first of them at query warm_first: the decisions before it stand (a decision depends on earlier
    // queries only), the exact sweeps go on from there on the previous `valid`.
    const bool warm = c->resolved && c->warm_first >= 0 && c->warm_first <= n && c->aln_verdicts && env_u32("IOC_RESOLVE_WARM", 1) == 1;
    const bool sharded = c->scored_sharded;
    if (sharded && (c->aln_verdicts || c->shard_world <= 1 || !c->shard_fn))
        return ioc_fail(c, IOC_ERR_STATE, "the scores are sharded (ioc_set_shard): ioc_resolve needs the same setting, without alignment verdicts");
    c->shard_exchanges = 0;
    auto exchange = [&](void* buf, int64_t count, int kind) -> int { return ioc_shard_exchange(c, buf, count, kind); };
    // initial guess (any guess converges to the same fixed point): "every query opens a cluster".
    // A guess from the all-pairs top Size (IOC_RESOLVE_GUESS=1) was measured SLOWER on config 2
    // (4 sweeps / 3.5 ms vs 3 sweeps / 2.0 ms): many entries with a large top still fail the mapped-ratio
    // test and do open clusters, and that side of the error cascades.
    if (n > 0 && !warm) {
        if (c->have_guess && env_u32("IOC_RESOLVE_GUESS", 0) == 1)
            HIPCHK(c, iock_guess_valid(s, n, c->d_off_fwd, c->d_off_rev, P<uint32_t>(c->b_top_all), P<uint8_t>(c->b_valid0)));
        else
            HIPCHK(c, hipMemsetAsync(c->b_valid0.p, 1, size_t(n), s));
    }
    if (!warm) c->cur_valid = 0;
    uint32_t* d_first_changed = P<uint32_t>(c->b_misc) + 8;
    unsigned long long* d_evals = reinterpret_cast<unsigned long long*>(P<uint8_t>(c->b_misc) + 128);
    HIPCHK(c, hipMemsetAsync(d_evals, 0, 8, s));
    DecideArgs a{};
    a.n = n;
    a.L = uint32_t(c->L);
    a.off_fwd = c->d_off_fwd;
    a.off_rev = c->d_off_rev;
    a.mins = c->d_min;
    a.pos = c->d_pos;
    a.hpc_len = c->d_hpc_len;
    a.err_cell = c->d_err_cell;
    a.min_total = c->d_min_total;
    a.left_err = P<uint8_t>(c->b_left_err);
    a.doff = P<int64_t>(c->b_doff);
    a.dvals = P<uint32_t>(c->b_dvals);
    a.dcount = P<uint32_t>(c->b_dcount);
    a.lset_off = P<int64_t>(c->b_lset_off);
    a.lset_val = P<uint32_t>(c->b_lset_val);
    a.cand_key = P<uint32_t>(c->b_cand_key);
    a.cand_size = P<uint32_t>(c->b_cand_size);
    a.cand_mapped = P<uint32_t>(c->b_cand_mapped);
    a.cand_count = P<uint32_t>(c->b_cand_count);
    a.glim = P<int32_t>(c->b_glim);
    a.dec_target = P<int32_t>(c->b_dec_target);
    a.dec_strand = P<int8_t>(c->b_dec_strand);
    a.flags = P<uint8_t>(c->b_flags);
    a.forced_t = P<int32_t>(c->b_forced_t);
    a.forced_s = P<int8_t>(c->b_forced_s);
    a.first_changed = d_first_changed;
    a.n_evals = d_evals;
    a.min_shared = c->params.min_shared;
    a.min_fraction = c->params.min_fraction;
    a.own_stride = sharded ? c->shard_world : 1;
    a.own_offset = sharded ? c->shard_rank : 0;
    // the upper bound of totalMapped per (query, strand, target error cell): ioc_score left it (ioc_gap_bounds_ready)
    a.gap_bound = (n > 0 && c->gap_bound_gen == c->query_gen && c->b_gap_bound.p) ? P<uint2>(c->b_gap_bound) : nullptr;
    if (c->aln_verdicts) {
        a.aln_t = P<int32_t>(c->b_aln_t);
        a.aln_s = P<int8_t>(c->b_aln_s);
        a.tie_count = P<uint32_t>(c->b_tie_count);
        a.tie_keys = P<uint32_t>(c->b_tie_keys);
    }
    // misc layout (uint32 words): [8] first_changed, [9] q_count (phase 1), [10] incomplete, [11] q_count (phase 2)
    const uint32_t q_cap = env_u32("IOC_QUEUE_CAP", 1u << 20);
    RESERVE(c, c->b_queue, size_t(q_cap) * 8);
    // [cut][top][done (bytes)][walk_n][walk_c]: per query
    const size_t walk_at = 2 * size_t(n) + (size_t(n) + 3) / 4 + 16;  // (words)
    RESERVE(c, c->b_cut, (walk_at + size_t(n) * (1 + IOC_WALK_SLOTS)) * 4 + 64);
    a.cut = P<int32_t>(c->b_cut);
    a.top = P<uint32_t>(c->b_cut) + n;
    a.done = reinterpret_cast<uint8_t*>(P<uint32_t>(c->b_cut) + 2 * size_t(n));
    a.walk_n = P<uint32_t>(c->b_cut) + walk_at;
    a.walk_c = a.walk_n + n;
    a.q_items = P<uint32_t>(c->b_queue);
    a.q_count = P<uint32_t>(c->b_misc) + 9;
    a.q_cap = q_cap;
    a.incomplete = P<uint32_t>(c->b_misc) + 10;
    const int eval_blocks = int(env_u32("IOC_EVAL_BLOCKS", 256 * 4));  // (what the chip holds: 4 workgroups of k_eval per CU)
    const bool diag = getenv("IOC_EVAL_DIAG") != nullptr;
    a.diag = nullptr;
    if (diag) {
        RESERVE(c, c->b_diag, 64);
        HIPCHK(c, hipMemsetAsync(c->b_diag.p, 0, 64, s));
        a.diag = P<unsigned long long>(c->b_diag);
    }
    // Two stages.  Lazy sweeps walk only each query's maximal-Size candidates and let a query whose top
    // candidates fail open a cluster provisionally; once those sweeps are stable, exact sweeps (the whole
    // walk) restart from query 0 on an almost final `valid`, so the long tails of the walk are evaluated
    // against actual clusters only.  The result is the fixed point of the exact sweeps either way.
    int first = 0, iters = 0, sweeps = 0;
    a.lazy = env_u32("IOC_RESOLVE_LAZY", 1) == 1 ? 1 : 0;
    if (warm) {
        a.lazy = 0;
        first = c->warm_first;
    }
    // The first exact sweep after the lazy fixed point would repeat, for every query and on the very `valid` the last lazy
    // sweeps ran on, the first half of a sweep (top, cut, the maximal-Size candidates: all cached): only its second half
    // runs — the rest of the walk of the queries the lazy sweeps let open a cluster provisionally (their `done` is 0).
    const bool skip_p1 = env_u32("IOC_RESOLVE_SKIP_P1", 1) == 1;
    bool p2only = false;
    while (n > 0) {
        if (first >= n) {
            // every query up to the last one is final for THIS stage: a lazy stage that ends on a change of the
            // last query still owes the exact sweeps (decisions of provisional cluster openers are not final)
            if (!a.lazy) break;
            a.lazy = 0;
            first = 0;
            p2only = skip_p1;
        }
        uint8_t* vin = c->cur_valid == 0 ? P<uint8_t>(c->b_valid0) : P<uint8_t>(c->b_valid1);
        uint8_t* vout = c->cur_valid == 0 ? P<uint8_t>(c->b_valid1) : P<uint8_t>(c->b_valid0);
        // (the control words are reset by the same launch that copies the final prefix of `valid`; they come back into pinned
        // memory: a pageable 16-byte upload and a pageable 12-byte read-back cost 45 us of host time per sweep between them)
        HIPCHK(c, iock_copy_prefix_valid(s, p2only ? n : first, vin, vout, d_first_changed));  // (second half only: `done` queries write nothing)
        a.first = first;
        a.valid_in = vin;
        a.valid_out = vout;
        if (p2only)
            HIPCHK(c, iock_decide_phase2(s, &a, n, eval_blocks, P<uint32_t>(c->b_misc) + 11));
        else
            HIPCHK(c, iock_decide_sweep(s, &a, n - first, eval_blocks, P<uint32_t>(c->b_misc) + 11));
        if (sharded) {
            // every rank wrote its own queries' share of valid_out and of the control words: what is not owned is zeroed, the
            // maximum over the ranks is the whole sweep's valid_out; first_changed is a minimum already and `incomplete` rides
            // the same all-reduce complemented (see k_shard_mask_u8)
            const int from = p2only ? 0 : first;
            HIPCHK(c, iock_shard_mask_u8(s, vout, nullptr, from, n, c->shard_world, c->shard_rank, d_first_changed));
            if (int rc = exchange(vout + from, int64_t(n - from), IOC_XCHG_MAX_U8)) return rc;
            if (int rc = exchange(d_first_changed, 3, IOC_XCHG_MIN_U32)) return rc;
        }
        volatile uint32_t* res = c->h_pin;
        // (control words at b_misc + 32 .. 44, the evaluation counter at b_misc + 128: one copy brings both, the counter of the
        // last sweep is the call's)
        HIPCHK(c, hipMemcpyAsync(c->h_pin, d_first_changed, 104, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        sweeps++;
        if (sweeps > 4 * n + 64) return ioc_fail(c, IOC_ERR_STATE, "resolve did not converge");
        const uint32_t res2 = sharded ? ~res[2] : res[2];
        if (res2 != 0) continue;  // work queue overflowed: same sweep again, the cache is fuller now
        p2only = false;
        iters++;
        c->cur_valid ^= 1;
        const uint32_t fc = res[0];
        if (fc == 0xFFFFFFFFu) {
            if (a.lazy) {  // lazy fixed point reached: switch to the exact sweeps
                a.lazy = 0;
                first = 0;
                p2only = skip_p1;
                continue;
            }
            break;  // fixed point: valid_out == valid_in
        }
        // queries <= fc are final: fc was computed from a correct prefix, everything before it
        // did not change ("fixed point of the greedy loop")
        first = int(fc) + 1;
    }
    if (sharded && n > 0) {
        // the decisions, gathered by owner (zero elsewhere: the sum of the words / the maximum of the bytes is the owner's value)
        HIPCHK(c, iock_shard_mask_i32(s, a.dec_target, n, c->shard_world, c->shard_rank));
        HIPCHK(c, iock_shard_mask_u8(s, reinterpret_cast<uint8_t*>(a.dec_strand), a.flags, 0, n, c->shard_world, c->shard_rank, nullptr));
        if (int rc = exchange(a.dec_target, n, IOC_XCHG_SUM_I32)) return rc;
        if (int rc = exchange(a.dec_strand, n, IOC_XCHG_MAX_U8)) return rc;
        if (int rc = exchange(a.flags, n, IOC_XCHG_MAX_U8)) return rc;
        HIPCHK(c, iock_shard_mask_i32(s, a.cut, n, c->shard_world, c->shard_rank));  // (ioc_get_cuts)
        if (int rc = exchange(a.cut, n, IOC_XCHG_SUM_I32)) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev[5], s));
    unsigned long long ev = 0;
    if (sweeps > 0) memcpy(&ev, const_cast<uint32_t*>(static_cast<volatile uint32_t*>(c->h_pin)) + 24, 8);
    if (sharded) HIPCHK(c, hipStreamSynchronize(s));  // (the gathers above)
    c->tm.n_mapped_evals = int64_t(ev);
    if (diag) {
        unsigned long long d[8];
        HIPCHK(c, hipMemcpy(d, c->b_diag.p, 64, hipMemcpyDeviceToHost));
fprintf(stderr, "[ioc eval diag] evals %llu: total %.0f cyc/eval = clear %.0f + insert %.0f + probe %.0f + gaps %.0f\n", d[2],
                d[2] ? double(d[1]) / d[2] : 0.0, d[2] ? double(d[5]) / d[2] : 0.0, d[2] ? double(d[6]) / d[2] : 0.0,
                d[2] ? double(d[7]) / d[2] : 0.0, d[2] ? double(d[0]) / d[2] : 0.0);
    }
    c->tm.resolve_iters = iters;
    if (n_iter) *n_iter = iters;
    c->resolved = true;
    c->warm_first = n;  // nothing has changed since this fixed point
    c->exp_valid = false;
    c->exp_dev = false;
    return IOC_OK;
}

int ioc_get_decisions(ioc_ctx* c, int32_t* target, int8_t* strand, uint8_t* flags)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    const size_t n = size_t(c->n);
    hipStream_t s = c->stream;
    if (n == 0) return IOC_OK;
    // (through pinned memory: three pageable read-backs cost 40 us of host time between them, once per resolve)
    const size_t need_b = n * 6 + 64;
    if (c->h_pin_big_cap < need_b) {
        if (c->h_pin_big) (void)hipHostFree(c->h_pin_big);
        c->h_pin_big = nullptr;
        c->h_pin_big_cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pin_big), need_b * 2, hipHostMallocDefault) == hipSuccess)
            c->h_pin_big_cap = need_b * 2;
        else
            (void)hipGetLastError();
    }
    if (!c->h_pin_big) {
        if (target) HIPCHK(c, hipMemcpyAsync(target, c->b_dec_target.p, n * 4, hipMemcpyDeviceToHost, s));
        if (strand) HIPCHK(c, hipMemcpyAsync(strand, c->b_dec_strand.p, n, hipMemcpyDeviceToHost, s));
        if (flags) HIPCHK(c, hipMemcpyAsync(flags, c->b_flags.p, n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return IOC_OK;
    }
    uint8_t* st = c->h_pin_big;
    if (target) HIPCHK(c, hipMemcpyAsync(st, c->b_dec_target.p, n * 4, hipMemcpyDeviceToHost, s));
    if (strand) HIPCHK(c, hipMemcpyAsync(st + n * 4, c->b_dec_strand.p, n, hipMemcpyDeviceToHost, s));
    if (flags) HIPCHK(c, hipMemcpyAsync(st + n * 5, c->b_flags.p, n, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (target) memcpy(target, st, n * 4);
    if (strand) memcpy(strand, st + n * 4, n);
    if (flags) memcpy(flags, st + n * 5, n);
    return IOC_OK;
}

int ioc_get_cuts(ioc_ctx* c, int32_t* cut)
{
    if (!c || !cut) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    const size_t n = size_t(c->n);
    if (n == 0) return IOC_OK;
    HIPCHK(c, hipMemcpyAsync(cut, c->b_cut.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_scored_candidates(ioc_ctx* c, int32_t q, int32_t cap, uint32_t* key, uint32_t* size)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->scored) return ioc_fail(c, IOC_ERR_STATE, "ioc_score first");
    if (q < 0 || q >= c->n || cap < 0 || (cap && (!key || !size))) return ioc_fail(c, IOC_ERR_ARG, "bad query index");
    const uint32_t L = uint32_t(c->L);
    const uint64_t cbase = 2ull * L * uint64_t(q) + uint64_t(q) * uint64_t(q > 0 ? q - 1 : 0);
    uint32_t cc = 0;
    HIPCHK(c, hipMemcpyAsync(&cc, P<uint32_t>(c->b_cand_count) + q, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t m = std::min<size_t>(cc, size_t(cap));
    if (m) {
        HIPCHK(c, hipMemcpyAsync(key, P<uint32_t>(c->b_cand_key) + cbase, m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(size, P<uint32_t>(c->b_cand_size) + cbase, m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return int(cc);
}

}  // extern "C"

// host copy of the per-query compaction thresholds (fast mode): a candidate that is missing from a query's list although its
// Size reaches the uniform `keep` was cut because it cannot pass any bound of totalMapped — exported as "rejected"
static int ensure_keep_host(ioc_ctx* c)
{
    if (!c->keep_q_on || !c->h_keep_q.empty() || c->n <= 0) return IOC_OK;
    c->h_keep_q.resize(size_t(c->n));
    HIPCHK(c, hipMemcpyAsync(c->h_keep_q.data(), c->b_keep_q.p, size_t(c->n) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

extern "C" {

int ioc_query_candidates(ioc_ctx* c, int32_t q, int32_t cap, int32_t* target, int8_t* strand, uint32_t* size,
                         uint32_t* first_index, uint32_t* total_mapped)
{
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    if (q < 0 || q >= c->n || cap < 0) return ioc_fail(c, IOC_ERR_ARG, "bad query index");
    hipStream_t s = c->stream;
    const uint32_t L = uint32_t(c->L);
    const uint32_t T = L + uint32_t(q);
    if (T == 0) return 0;
    {
        const int rk = ensure_keep_host(c);
        if (rk != IOC_OK) return rk;
    }
    const uint32_t cut_below = c->keep_q_on ? c->h_keep_q[size_t(q)] : 0u;
    // (called once per tied / order-dependent query of a round, thousands of times on a large batch: scratch
    // buffers are kept in the context and everything comes back with one synchronisation — the candidate list
    // is copied at its capacity, its fill count arrives with it)
    const auto tq0 = std::chrono::steady_clock::now();
    RESERVE(c, c->b_qhist, size_t(2) * T * 4);
    RESERVE(c, c->b_qfirst, size_t(2) * T * 4);
    const uint8_t* valid = c->cur_valid == 0 ? P<uint8_t>(c->b_valid0) : P<uint8_t>(c->b_valid1);
    // the table is compacted on the device: a query hits a few dozen of its 2 T possible (target, strand) cells
    const uint32_t n2 = 2u * T, QC = std::min<uint32_t>(n2, 32768u);  // (a query of a 30 k-read batch hits ~20 k cells)
    RESERVE(c, c->b_qout, (size_t(1) + 3 * size_t(QC)) * 4);
    std::vector<uint32_t> ho(size_t(1) + 3 * size_t(QC));
    uint32_t cc = 0;
    const uint64_t cbase = 2ull * L * uint64_t(q) + uint64_t(q) * uint64_t(q > 0 ? q - 1 : 0);
    const size_t ccap = size_t(2) * T, cfirst = std::min<size_t>(ccap, 4096);  // this query's candidate list: capacity, first copy
    std::vector<uint32_t> ck(cfirst), cm(cfirst);
    hipError_t e = hipMemsetAsync(c->b_qhist.p, 0, size_t(n2) * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(c->b_qfirst.p, 0xFF, size_t(n2) * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(c->b_qout.p, 0, 4, s);
    if (e == hipSuccess)
        e = iock_query_table(s, q, L, c->d_off_fwd, c->d_off_rev, c->d_min, c->b_rows.p, c->cap, hash_shift(c->cap),
                             c->b_post.p, valid, P<uint32_t>(c->b_qhist), P<uint32_t>(c->b_qfirst), c->post16);
    if (e == hipSuccess) e = iock_query_compact(s, P<uint32_t>(c->b_qhist), P<uint32_t>(c->b_qfirst), n2, QC, P<uint32_t>(c->b_qout));
    if (e == hipSuccess) e = hipMemcpyAsync(ho.data(), c->b_qout.p, ho.size() * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&cc, P<uint32_t>(c->b_cand_count) + q, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ck.data(), P<uint32_t>(c->b_cand_key) + cbase, cfirst * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(cm.data(), P<uint32_t>(c->b_cand_mapped) + cbase, cfirst * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, std::string("ioc_query_candidates: ") + hipGetErrorString(e));
    if (getenv("IOC_TRACE_Q"))
        fprintf(stderr, "[ioc] candidate table of query %d: %.3f ms on the device path, %u cells, %u candidates\n", q,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count(), ho[0], cc);
    if (cc > ccap) return ioc_fail(c, IOC_ERR_STATE, "candidate list longer than its capacity");
    if (cc > cfirst) {  // (a very long candidate list: the rest of it)
        ck.resize(cc);
        cm.resize(cc);
        e = hipMemcpyAsync(ck.data() + cfirst, P<uint32_t>(c->b_cand_key) + cbase + cfirst, (cc - cfirst) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(cm.data() + cfirst, P<uint32_t>(c->b_cand_mapped) + cbase + cfirst, (cc - cfirst) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, std::string("ioc_query_candidates: ") + hipGetErrorString(e));
    }
    struct Hit {
        uint32_t idx, size, first;
    };
    std::vector<Hit> hits;
    if (ho[0] <= QC) {
        hits.resize(ho[0]);
        for (uint32_t i = 0; i < ho[0]; ++i) hits[i] = Hit{ho[1 + 3 * i], ho[2 + 3 * i], ho[3 + 3 * i]};
        std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.idx < b.idx; });
    } else {  // more hit cells than the compact buffer holds: the whole table
        std::vector<uint32_t> hh(n2), hf(n2);
        e = hipMemcpyAsync(hh.data(), c->b_qhist.p, hh.size() * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(hf.data(), c->b_qfirst.p, hf.size() * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, std::string("ioc_query_candidates: ") + hipGetErrorString(e));
        for (uint32_t i = 0; i < n2; ++i)
            if (hh[i]) hits.push_back(Hit{i, hh[i], hf[i]});
    }
    // cached totalMapped values of this query's candidate list, by table cell
    std::vector<std::pair<uint32_t, uint32_t>> mapped;
    mapped.reserve(cc);
    for (uint32_t i = 0; i < cc; ++i) {
        const uint32_t tg = ck[i] >> 1, sb = ck[i] & 1u;
        if (tg < T) mapped.emplace_back(sb * T + tg, cm[i]);
    }
    std::stable_sort(mapped.begin(), mapped.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
    int out = 0;
    for (const Hit& h : hits) {  // ascending cell index = (strand +1 first, then -1), targets ascending
        if (out < cap) {
            const uint32_t sb = h.idx >= T ? 1u : 0u, t = h.idx - sb * T;
            if (target) target[out] = int32_t(t);
            if (strand) strand[out] = sb ? -1 : 1;
            if (size) size[out] = h.size;
            if (first_index) first_index[out] = h.first;
            if (total_mapped) {
                auto it = std::lower_bound(mapped.begin(), mapped.end(), std::make_pair(h.idx, 0u),
                                           [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
                // (the last entry of a cell wins, as the dense table's overwrite did)
                uint32_t mv = h.size < cut_below ? 0xFFFFFFFEu : 0xFFFFFFFFu;  // (cut from the list: rejected by the bound)
                for (; it != mapped.end() && it->first == h.idx; ++it) mv = it->second;
                total_mapped[out] = mv;
            }
        }
        out++;
    }
    if (out > cap) return ioc_fail(c, IOC_ERR_CAPACITY, "candidate buffer too small: need " + std::to_string(out));
    return out;
}

}  // extern "C"

int ioc_query_candidates_many(ioc_ctx* c, const std::vector<int>& qs, std::vector<IocCandTable>& out)
{
    out.clear();
    if (!c) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    for (int q : qs)
        if (q < 0 || q >= c->n) return ioc_fail(c, IOC_ERR_ARG, "bad query index");
    out.resize(qs.size());
    if (qs.empty()) return IOC_OK;
    {
        const int rk = ensure_keep_host(c);
        if (rk != IOC_OK) return rk;
    }
    hipStream_t s = c->stream;
    const uint32_t L = uint32_t(c->L);
    const size_t n = size_t(c->n);
    // the candidate lists' fill counts, once
    std::vector<uint32_t> ccount(n);
    HIPCHK(c, hipMemcpyAsync(ccount.data(), c->b_cand_count.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const uint8_t* valid = c->cur_valid == 0 ? P<uint8_t>(c->b_valid0) : P<uint8_t>(c->b_valid1);
    size_t done = 0;
    while (done < qs.size()) {
        // a chunk: tables of `stride` words per query (the largest 2 T of the chunk), at most 256 MB each
        uint32_t maxT = 1;
        size_t cnt = 0;
        while (done + cnt < qs.size() && cnt < 512) {
            const uint32_t T = L + uint32_t(qs[done + cnt]);
            const uint32_t mt = std::max(maxT, T);
            if (cnt > 0 && uint64_t(cnt + 1) * 2ull * mt * 4ull > (256ull << 20)) break;
            maxT = mt;
            ++cnt;
        }
        const uint64_t stride = 2ull * maxT;
        const uint32_t QC = uint32_t(stride);  // (every cell could be hit; only the filled part of a slice is copied back)
        const size_t out_words = size_t(1) + 3 * size_t(QC);
        RESERVE(c, c->b_qhist, size_t(cnt) * stride * 4);
        RESERVE(c, c->b_qfirst, size_t(cnt) * stride * 4);
        RESERVE(c, c->b_qout, size_t(cnt) * out_words * 4);
        RESERVE(c, c->b_qlist, size_t(cnt) * 4);
        std::vector<int32_t> ql(qs.begin() + done, qs.begin() + done + cnt);
        const bool trq = getenv("IOC_TRACE") != nullptr;
        auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tq0 = tnow();
        HIPCHK(c, hipMemcpyAsync(c->b_qlist.p, ql.data(), cnt * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemsetAsync(c->b_qhist.p, 0, size_t(cnt) * stride * 4, s));
        HIPCHK(c, hipMemsetAsync(c->b_qfirst.p, 0xFF, size_t(cnt) * stride * 4, s));
        HIPCHK(c, hipMemsetAsync(c->b_qout.p, 0, size_t(cnt) * out_words * 4, s));
        HIPCHK(c, iock_query_table_many(s, int(cnt), P<int32_t>(c->b_qlist), stride, L, c->d_off_fwd, c->d_off_rev, c->d_min, c->b_rows.p,
                                        c->cap, hash_shift(c->cap), c->b_post.p, valid, P<uint32_t>(c->b_qhist), P<uint32_t>(c->b_qfirst),
                                        c->post16, QC, P<uint32_t>(c->b_qout)));
        if (trq) HIPCHK(c, hipStreamSynchronize(s));
        const double tq1 = tnow();
        // the fill counts first (one strided copy), then the filled part of every slice
        std::vector<uint32_t> hcnt(cnt);
        HIPCHK(c, hipMemcpy2DAsync(hcnt.data(), 4, c->b_qout.p, out_words * 4, 4, cnt, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        std::vector<size_t> hoff(cnt + 1, 0);
        for (size_t x = 0; x < cnt; ++x) hoff[x + 1] = hoff[x] + 1 + 3 * size_t(std::min<uint32_t>(hcnt[x], QC));
        std::vector<uint32_t> ho(hoff[cnt]);
        for (size_t x = 0; x < cnt; ++x)
            HIPCHK(c, hipMemcpyAsync(ho.data() + hoff[x], P<uint32_t>(c->b_qout) + x * out_words, (hoff[x + 1] - hoff[x]) * 4, hipMemcpyDeviceToHost, s));
        if (trq) HIPCHK(c, hipStreamSynchronize(s));
        const double tq2 = tnow();
        // the queries' candidate lists (cached totalMapped values)
        std::vector<std::vector<uint32_t>> ck(cnt), cm(cnt);
        for (size_t x = 0; x < cnt; ++x) {
            const int q = ql[x];
            const uint32_t cc = ccount[size_t(q)];
            const uint64_t cbase = 2ull * L * uint64_t(q) + uint64_t(q) * uint64_t(q > 0 ? q - 1 : 0);
            if (cc > 2u * (L + uint32_t(q))) return ioc_fail(c, IOC_ERR_STATE, "candidate list longer than its capacity");
            ck[x].resize(cc);
            cm[x].resize(cc);
            if (cc) {
                HIPCHK(c, hipMemcpyAsync(ck[x].data(), P<uint32_t>(c->b_cand_key) + cbase, size_t(cc) * 4, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipMemcpyAsync(cm[x].data(), P<uint32_t>(c->b_cand_mapped) + cbase, size_t(cc) * 4, hipMemcpyDeviceToHost, s));
            }
        }
        HIPCHK(c, hipStreamSynchronize(s));
        if (trq)
            fprintf(stderr, "[ioc]   %zu candidate tables: kernels %.1f ms, compact lists back %.1f ms, candidate lists back %.1f ms\n", cnt, tq1 - tq0,
                    tq2 - tq1, tnow() - tq2);
        std::vector<int> overflow(cnt, 0);
        const double tq3 = tnow();
        ioc_parallel_for(cnt, [&](size_t x) {
            const int q = ql[x];
            const uint32_t T = L + uint32_t(q);
            const uint32_t* o = ho.data() + hoff[x];
            if (o[0] > QC) {
                overflow[x] = 1;
                return;
            }
            struct Hit {
                uint32_t idx, size, first;
            };
            std::vector<Hit> hits(o[0]);
            for (uint32_t i = 0; i < o[0]; ++i) hits[i] = Hit{o[1 + 3 * i], o[2 + 3 * i], o[3 + 3 * i]};
            std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.idx < b.idx; });
            std::vector<std::pair<uint32_t, uint32_t>> mapped;
            mapped.reserve(ck[x].size());
            for (size_t i = 0; i < ck[x].size(); ++i) {
                const uint32_t tg = ck[x][i] >> 1, sb = ck[x][i] & 1u;
                if (tg < T) mapped.emplace_back(sb * T + tg, cm[x][i]);
            }
            auto by_cell = [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; };
            std::stable_sort(mapped.begin(), mapped.end(), by_cell);
            IocCandTable& t = out[done + x];
            t.q = q;
            t.tg.resize(hits.size());
            t.st.resize(hits.size());
            t.sz.resize(hits.size());
            t.fi.resize(hits.size());
            t.tm.resize(hits.size());
            for (size_t i = 0; i < hits.size(); ++i) {
                const uint32_t sb = hits[i].idx >= T ? 1u : 0u;
                t.tg[i] = int32_t(hits[i].idx - sb * T);
                t.st[i] = sb ? -1 : 1;
                t.sz[i] = hits[i].size;
                t.fi[i] = hits[i].first;
                auto it = std::lower_bound(mapped.begin(), mapped.end(), std::make_pair(hits[i].idx, 0u), by_cell);
                uint32_t mv = (c->keep_q_on && hits[i].size < c->h_keep_q[size_t(q)]) ? 0xFFFFFFFEu : 0xFFFFFFFFu;  // (cut from the list: rejected by the bound)
                for (; it != mapped.end() && it->first == hits[i].idx; ++it) mv = it->second;
                t.tm[i] = mv;
            }
        });
        if (trq) fprintf(stderr, "[ioc]   ... lists sorted on the host's cores in %.1f ms\n", tnow() - tq3);
        // (a table with more cells than the compact buffer holds: the one-query path copies it whole)
        for (size_t x = 0; x < cnt; ++x)
            if (overflow[x]) {
                const int q = ql[x];
                const size_t cap = size_t(2) * (L + uint32_t(q)) + 1;
                IocCandTable& t = out[done + x];
                t.q = q;
                t.tg.resize(cap);
                t.st.resize(cap);
                t.sz.resize(cap);
                t.fi.resize(cap);
                t.tm.resize(cap);
                const int nc = ioc_query_candidates(c, q, int32_t(cap - 1), t.tg.data(), t.st.data(), t.sz.data(), t.fi.data(), t.tm.data());
                if (nc < 0) return nc;
                t.tg.resize(size_t(nc));
                t.st.resize(size_t(nc));
                t.sz.resize(size_t(nc));
                t.fi.resize(size_t(nc));
                t.tm.resize(size_t(nc));
            }
        done += cnt;
    }
    return IOC_OK;
}

extern "C" {

// The export (device -> host copy of the combined index, renumbering, sort by key) is computed once per
// resolve and kept: callers size with a first call (keys == NULL) and fetch with a second one.
static int index_export_compute(ioc_ctx* c)
{
    // The final MinDB = the index's posting lists restricted to the targets that are clusters, with final ids
    // (AddMinimizers for every query that opened a cluster, minimizer.cpp:31-42).  Filtering and renumbering run on the
    // device (k_export_count / k_export_fill: one wave per list); the host only orders the keys (the reference's
    // unordered_map has no order of its own: the CSR is given in ascending key order) and takes the compact result.
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t nslots = c->cap + 1;
    const bool tr = getenv("IOC_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) {
        if (tr) {
            const double t1 = now();
            fprintf(stderr, "[ioc]   export: %-40s %8.3f ms\n", what, t1 - t0);
            t0 = t1;
        }
    };
    std::vector<uint8_t> valid(size_t(c->n) + 1);
    const void* v = c->cur_valid == 0 ? c->b_valid0.p : c->b_valid1.p;
    if (c->n) HIPCHK(c, hipMemcpyAsync(valid.data(), v, size_t(c->n), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    lap("valid to the host");
    // final cluster id of query i that opened a cluster = L + rank among such queries (cluster.cpp:178)
    std::vector<int32_t> cid(size_t(c->n) + 1, -1);
    int32_t next = c->L;
    for (int i = 0; i < c->n; ++i)
        if (valid[size_t(i)]) cid[size_t(i)] = next++;
    RESERVE(c, c->b_exp_cid, (size_t(c->n) + 1) * 4);
    RESERVE(c, c->b_exp_cnt, size_t(nslots) * 4);
    RESERVE(c, c->b_exp_off, size_t(nslots) * 8);
    HIPCHK(c, hipMemcpyAsync(c->b_exp_cid.p, cid.data(), (size_t(c->n) + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, iock_export_count(s, nslots, P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt), c->b_post.p, c->post16, uint32_t(c->L),
                                P<int32_t>(c->b_exp_cid), P<uint32_t>(c->b_exp_cnt)));
    // keys whose every contributor joined another cluster were never inserted by AddMinimizers: no entry (the reference
    // keeps keys with emptied lists only through UpdateMinDB, i.e. in consensus mode)
    if (env_u32("IOC_EXPORT_HOST_ORDER", 0) == 0) {
        // the kept keys in ascending order, the offsets of their lists and every slot's place: on the device (ioc_sort.hip)
        auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
        const size_t tmpb = iock_export_order_temp(nslots);
        const size_t o_k0 = 0, o_k1 = o_k0 + up(size_t(nslots) * 8), o_v0 = o_k1 + up(size_t(nslots) * 8), o_v1 = o_v0 + up(size_t(nslots) * 4),
                     o_sc = o_v1 + up(size_t(nslots) * 4), o_so = o_sc + up((size_t(nslots) + 1) * 8), o_ok = o_so + up((size_t(nslots) + 1) * 8),
                     o_nr = o_ok + up(size_t(nslots) * 4), o_tmp = o_nr + 256;
        RESERVE(c, c->b_exp_work, o_tmp + tmpb);
        uint8_t* wk = P<uint8_t>(c->b_exp_work);
        unsigned long long* d_soff = reinterpret_cast<unsigned long long*>(wk + o_so);
        uint32_t* d_nrows = reinterpret_cast<uint32_t*>(wk + o_nr);
        HIPCHK(c, iock_export_order(s, nslots, c->cap, P<uint32_t>(c->b_keys), P<uint32_t>(c->b_exp_cnt),
                                    reinterpret_cast<unsigned long long*>(wk + o_k0), reinterpret_cast<unsigned long long*>(wk + o_k1),
                                    reinterpret_cast<uint32_t*>(wk + o_v0), reinterpret_cast<uint32_t*>(wk + o_v1),
                                    reinterpret_cast<unsigned long long*>(wk + o_sc), d_soff, wk + o_tmp, tmpb, d_nrows,
                                    reinterpret_cast<uint32_t*>(wk + o_ok), P<int64_t>(c->b_exp_off)));
        uint32_t nrows = 0;
        unsigned long long total = 0;
        HIPCHK(c, hipMemcpyAsync(&nrows, d_nrows, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&total, d_soff + nslots, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        lap("keys ordered on the device");
        if (total > 0) {
            RESERVE(c, c->b_exp_out, size_t(total) * 4);
            HIPCHK(c, iock_export_fill(s, nslots, P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt), c->b_post.p, c->post16, uint32_t(c->L),
                                       P<int32_t>(c->b_exp_cid), P<uint32_t>(c->b_exp_cnt), P<int64_t>(c->b_exp_off), P<uint32_t>(c->b_exp_out)));
        }
        // (the result stays on the device: ioc_index_export copies it straight into the caller's arrays)
        c->exp_nrows = nrows;
        c->exp_total = total;
        c->exp_o_keys = o_ok;
        c->exp_o_offs = o_so;  // (soff[nrows] = total)
        c->exp_dev = true;
        return IOC_OK;
    }
    std::vector<uint32_t> hk(nslots), hcnt(nslots);  // (IOC_EXPORT_HOST_ORDER=1: the keys ordered by the host, for comparison)
    HIPCHK(c, hipMemcpyAsync(hk.data(), c->b_keys.p, size_t(nslots) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(hcnt.data(), c->b_exp_cnt.p, size_t(nslots) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::vector<uint64_t> rows;  // key << 32 | slot
    rows.reserve(nslots / 2);
    for (uint32_t sl = 0; sl < nslots; ++sl)
        if (hcnt[sl]) rows.push_back((uint64_t(sl == c->cap ? 0xFFFFFFFFu : hk[sl]) << 32) | sl);
    std::sort(rows.begin(), rows.end());
    std::vector<int64_t> hoff(nslots, 0);
    c->exp_keys.resize(rows.size());
    c->exp_offs.resize(rows.size() + 1);
    int64_t tot = 0;
    for (size_t i = 0; i < rows.size(); ++i) {
        const uint32_t sl = uint32_t(rows[i] & 0xFFFFFFFFu);
        c->exp_keys[i] = uint32_t(rows[i] >> 32);
        c->exp_offs[i] = tot;
        hoff[sl] = tot;
        tot += int64_t(hcnt[sl]);
    }
    c->exp_offs[rows.size()] = tot;
    c->exp_post.resize(size_t(tot));
    lap("keys ordered, offsets");
    if (tot > 0) {
        RESERVE(c, c->b_exp_out, size_t(tot) * 4);
        HIPCHK(c, hipMemcpyAsync(c->b_exp_off.p, hoff.data(), size_t(nslots) * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, iock_export_fill(s, nslots, P<uint32_t>(c->b_off), P<uint32_t>(c->b_cnt), c->b_post.p, c->post16, uint32_t(c->L),
                                   P<int32_t>(c->b_exp_cid), P<uint32_t>(c->b_exp_cnt), P<int64_t>(c->b_exp_off), P<uint32_t>(c->b_exp_out)));
        HIPCHK(c, hipMemcpyAsync(c->exp_post.data(), c->b_exp_out.p, size_t(tot) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    lap("fill kernel, postings to the host");
    c->exp_valid = true;
    return IOC_OK;
}

int ioc_index_export(ioc_ctx* c, int64_t* n_keys, int64_t* n_postings, uint32_t* keys, int64_t* offs,
                     uint32_t* postings)
{
    if (!c) return IOC_ERR_ARG;
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    if (!c->exp_valid && !c->exp_dev) {
        int r = index_export_compute(c);
        if (r != IOC_OK) return r;
    }
    if (!c->exp_valid) {  // on the device: sizes, and — with arrays — three copies into the caller's memory
        HIPCHK(c, hipSetDevice(c->device));
        hipStream_t s = c->stream;
        const double t0 = getenv("IOC_TRACE") ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        const uint8_t* wk = P<uint8_t>(c->b_exp_work);
        if (keys && c->exp_nrows) HIPCHK(c, hipMemcpyAsync(keys, wk + c->exp_o_keys, size_t(c->exp_nrows) * 4, hipMemcpyDeviceToHost, s));
        if (offs) HIPCHK(c, hipMemcpyAsync(offs, wk + c->exp_o_offs, (size_t(c->exp_nrows) + 1) * 8, hipMemcpyDeviceToHost, s));
        if (postings && c->exp_total) HIPCHK(c, hipMemcpyAsync(postings, c->b_exp_out.p, size_t(c->exp_total) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (getenv("IOC_TRACE") && (keys || offs || postings))
            fprintf(stderr, "[ioc]   export: keys / offsets / postings to the caller's arrays %8.3f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
        if (n_keys) *n_keys = int64_t(c->exp_nrows);
        if (n_postings) *n_postings = int64_t(c->exp_total);
        return IOC_OK;
    }
    const size_t nk = c->exp_keys.size(), np = c->exp_post.size();
    if (keys && nk) memcpy(keys, c->exp_keys.data(), nk * 4);
    if (offs) memcpy(offs, c->exp_offs.data(), (nk + 1) * 8);
    if (postings && np) memcpy(postings, c->exp_post.data(), np * 4);
    if (n_keys) *n_keys = int64_t(nk);
    if (n_postings) *n_postings = int64_t(np);
    return IOC_OK;
}

int ioc_count_reference_postings(ioc_ctx* c, int64_t* n_postings)
{
    if (!c || !n_postings) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resolved) return ioc_fail(c, IOC_ERR_STATE, "ioc_resolve first");
    hipStream_t s = c->stream;
    RESERVE(c, c->b_misc, 256);
    unsigned long long* d_sum = reinterpret_cast<unsigned long long*>(P<uint8_t>(c->b_misc) + 192);
    HIPCHK(c, hipMemsetAsync(d_sum, 0, 8, s));
    const uint8_t* valid = c->cur_valid == 0 ? P<uint8_t>(c->b_valid0) : P<uint8_t>(c->b_valid1);
    const uint32_t range = env_u32("IOC_SCORE_RANGE", 8192);
    iock_set_score_oob(c->score_oob);
    HIPCHK(c, iock_score(s, c->n, uint32_t(c->L), c->d_off_fwd, c->d_off_rev, c->d_min, c->b_rows.p, c->cap,
                         hash_shift(c->cap), c->b_post.p, range, uint32_t(c->keep),
                         P<uint32_t>(c->b_cand_key), P<uint32_t>(c->b_cand_size), P<uint32_t>(c->b_cand_count),
                         nullptr, valid, d_sum, P<uint32_t>(c->b_part), nullptr, c->post16, P<uint32_t>(c->b_pmins),
                         P<uint32_t>(c->b_pbnd)));
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *n_postings = int64_t(h);
    return IOC_OK;
}

int ioc_get_timings(ioc_ctx* c, ioc_timings* out)
{
    if (!c || !out) return IOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    c->tm.score_oob = c->score_oob;
    c->tm.score_oob_probe = c->score_oob_probe;
    if (c->built && c->n_post < 0 && c->b_off.p) {  // (the sorted build does not wait for its postings count: it lands in pinned memory)
        const uint32_t t = static_cast<volatile uint32_t*>(c->h_pin)[40];
        c->n_post = t;
        c->tm.n_index_postings = t;
    }
    if (c->built && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->tm.ms_build = ms;
    if (c->scored && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->tm.ms_score = ms;
    if (c->resolved && hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) c->tm.ms_resolve = ms;
    if (c->scored && getenv("IOC_COUNT_CANDIDATES")) {
        std::vector<uint32_t> cc(size_t(c->n) + 1);
        if (c->n) HIPCHK(c, hipMemcpy(cc.data(), c->b_cand_count.p, size_t(c->n) * 4, hipMemcpyDeviceToHost));
        int64_t t = 0;
        for (int i = 0; i < c->n; ++i) t += cc[size_t(i)];
        c->tm.n_candidates = t;
    }
    *out = c->tm;
    return IOC_OK;
}

}  // extern "C"
