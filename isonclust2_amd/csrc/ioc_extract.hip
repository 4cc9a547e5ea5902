// ioc_extract.hip — sort-stage feeders of the path on the GPU.
//
//   k_qual_scores    CalcQualScore + CalcErrorRate          src/qualscore.cpp:107-154
//   k_hpc            HomopolymerCompress (+ base check)     src/hpc.cpp:4-32, src/util.cpp:13-38
//   k_seq_error      CalcErrorRate over HPC qualities       src/qualscore.cpp:78-80, 147-154
//   k_minimizers     KmerEncodeSeq + RevComp + GetKmerMinimizers, both strands
//                                                           src/kmer_index.cpp:5-17, src/minimizer.cpp:78-123
//
// Floating point: the two fp64 recurrences (running product of the quality score, running sum of
// the error rate) are evaluated in exactly the reference's operation order by ONE logical thread per
// read; the independent parts (table lookups, the p_enter / p_leave quotient) are computed 64 lanes
// wide and handed to the serial chain with readlane.  Compiled with -ffp-contract=off: no FMA, like
// the reference build (-msse3, CMakeLists.txt:45).  fp64 '/' is IEEE-correctly rounded on gfx950.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ioc_internal.h"
#include "ioc_kernels.h"

#define XB 256
#define XW (XB / 64)

namespace {

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & 63; }
__device__ __forceinline__ int wave_id() { return int(threadIdx.x) >> 6; }

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = __builtin_amdgcn_readlane(unsigned(u), l);
    unsigned hi = __builtin_amdgcn_readlane(unsigned(u >> 32), l);
    return __longlong_as_double((unsigned long long)hi << 32 | lo);
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t& total, uint32_t* sh)
{
    uint32_t incl = wave_incl_scan_u32(v);
    if (lane_id() == 63) sh[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < XW; ++w) {
        uint32_t s = sh[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// One wave per read.  score: sum over k-windows of prod(1 - p_err) with cur *= (p_enter / p_leave);
// err: mean of the uncapped table.  Reads with len <= 2k get (-1, 1.0) (qualscore.cpp:22-34);
// score <= 0 becomes -1.  A byte > 128 makes the reference's vector::at throw: reported as NaN.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(XB)
k_qual_scores(int n, const int64_t* __restrict__ offs, const uint8_t* __restrict__ qual, int k,
              const double* __restrict__ tab_capped, const double* __restrict__ tab_nomin,
              double* __restrict__ score, double* __restrict__ err)
{
    const int r = (blockIdx.x * XB + threadIdx.x) >> 6;
    if (r >= n) return;
    const int lane = lane_id();
    const int64_t b = offs[r];
    const int64_t len = offs[r + 1] - b;
    if (len <= 2 * int64_t(k)) {
        if (lane == 0) {
            score[r] = -1.0;
            err[r] = 1.0;
        }
        return;
    }
    const uint8_t* q = qual + b;
    bool bad = false;
    // product over the first k qualities, left to right
    double cur = 1.0;
    double psum = 0.0;
    for (int64_t c0 = 0; c0 < k; c0 += 64) {
        const int64_t i = c0 + lane;
        double pe = 1.0, pn = 0.0;
        if (i < k) {
            const uint32_t ch = q[i];
            bad |= ch > 128;
            pe = 1.0 - tab_capped[ch > 128 ? 0 : ch];
            pn = tab_nomin[ch > 128 ? 0 : ch];
        }
        const int m = int((k - c0 < 64) ? (k - c0) : 64);
        for (int l = 0; l < m; ++l) {
            cur *= readlane_f64(pe, l);
            psum += readlane_f64(pn, l);
        }
    }
    double sum = cur;
    for (int64_t c0 = k; c0 < len; c0 += 64) {
        const int64_t i = c0 + lane;
        double ratio = 1.0, pn = 0.0;
        if (i < len) {
            const uint32_t ce = q[i], cl = q[i - k];
            bad |= ce > 128;
            const double pe = 1.0 - tab_capped[ce > 128 ? 0 : ce];
            const double pl = 1.0 - tab_capped[cl > 128 ? 0 : cl];
            ratio = pe / pl;
            pn = tab_nomin[ce > 128 ? 0 : ce];
        }
        const int m = int((len - c0 < 64) ? (len - c0) : 64);
        for (int l = 0; l < m; ++l) {
            cur *= readlane_f64(ratio, l);
            sum += cur;
            psum += readlane_f64(pn, l);
        }
    }
    const bool anybad = __ballot(bad) != 0;
    if (lane == 0) {
        double qs = sum;
        if (qs <= 0) qs = -1.0;
        score[r] = anybad ? nan("") : qs;
        err[r] = anybad ? nan("") : psum / double(len);
    }
}

// ---------------------------------------------------------------------------------------------------
// HPC: one workgroup per read.  hseq = base codes 0..3 (A C G T) of the run heads, hqual = max quality
// character of each run.  status: 0 ok, 2 non-ACGT base.  hseq/hqual are written at the read's own
// offset (capacity = raw length).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint8_t c)
{
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

__global__ void __launch_bounds__(XB)
k_hpc(int n, const int64_t* __restrict__ offs, const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
      uint8_t* __restrict__ hseq, uint8_t* __restrict__ hqual, uint32_t* __restrict__ hlen,
      int32_t* __restrict__ status)
{
    __shared__ uint32_t sh[XW];
    __shared__ uint32_t s_bad;
    const int r = blockIdx.x;
    if (r >= n) return;
    const int64_t b = offs[r];
    const int64_t len = offs[r + 1] - b;
    const uint8_t* s = seq + b;
    const uint8_t* q = qual + b;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    uint32_t base = 0;
    bool bad = false;
    for (int64_t c0 = 0; c0 < len; c0 += XB) {
        const int64_t i = c0 + threadIdx.x;
        uint32_t flag = 0;
        uint8_t ch = 0;
        if (i < len) {
            ch = s[i];
            flag = (i == 0) || (s[i - 1] != ch);
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(flag, tot, sh);
        if (flag) {
            const uint32_t code = base_code(ch);
            bad |= code > 3;
            // maximum quality of the run [i, end of run)
            uint8_t mq = q[i];
            for (int64_t t = i + 1; t < len && s[t] == ch; ++t) mq = q[t] > mq ? q[t] : mq;
            hseq[b + base + ex] = uint8_t(code);
            hqual[b + base + ex] = mq;
        }
        base += tot;
    }
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        hlen[r] = base;
        status[r] = s_bad ? 2 : 0;
    }
}

// mean of tab[q] over a sequence, summed in order (one wave per read)
__global__ void __launch_bounds__(XB)
k_seq_error(int n, const int64_t* __restrict__ offs, const uint8_t* __restrict__ hqual,
            const uint32_t* __restrict__ hlen, const double* __restrict__ tab, double* __restrict__ err)
{
    const int r = (blockIdx.x * XB + threadIdx.x) >> 6;
    if (r >= n) return;
    const int lane = lane_id();
    const uint8_t* q = hqual + offs[r];
    const int64_t len = hlen[r];
    double psum = 0.0;
    for (int64_t c0 = 0; c0 < len; c0 += 64) {
        const int64_t i = c0 + lane;
        double pn = 0.0;
        if (i < len) {
            const uint32_t ch = q[i];
            pn = tab[ch > 128 ? 0 : ch];
        }
        const int m = int((len - c0 < 64) ? (len - c0) : 64);
        for (int l = 0; l < m; ++l) psum += readlane_f64(pn, l);
    }
    if (lane == 0) err[r] = len ? psum / double(len) : 1.0;
}

// ---------------------------------------------------------------------------------------------------
// Minimizers.  One workgroup per (read, strand).  The HPC read is staged 2-bit-packed in LDS
// (coalesced global reads, 16 bases per word); the reverse strand is read through the same image
// (base p of the reverse complement = 3 - code[L-1-p]).  k-mer value = the reference's 32-bit
// unsigned polynomial (for k > 16 only the last 16 bases survive the wrap).  Window j covers k-mers
// j .. j+W-1 (W = w-k+1); its minimizer is the LEFTMOST minimum; a minimizer is emitted when the
// argmin differs from that of window j-1 — equivalent to the reference's sliding rule
// (tests/test_oracle_golden.py::test_minimizers_equal_argmin_change_formulation).
// pass 0 counts, pass 1 writes (min, pos) at the read's offset; Index is the ordinal.
// ---------------------------------------------------------------------------------------------------
#define XTILE 256
__device__ __forceinline__ uint32_t code_at(const uint32_t* packed, uint32_t L, bool rev, uint32_t p)
{
    const uint32_t src = rev ? (L - 1 - p) : p;
    const uint32_t c = (packed[src >> 4] >> ((src & 15u) * 2)) & 3u;
    return rev ? 3u - c : c;
}

__global__ void __launch_bounds__(XB)
k_minimizers(int n, const int64_t* __restrict__ offs, const uint8_t* __restrict__ hseq,
             const uint32_t* __restrict__ hlen, const int32_t* __restrict__ status, int k, int w, int pass,
             const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_rev,
             uint32_t* __restrict__ cnt_fwd, uint32_t* __restrict__ cnt_rev, uint32_t* __restrict__ omin,
             uint32_t* __restrict__ opos, uint32_t lds_words)
{
    extern __shared__ uint32_t packed[];  // lds_words
    __shared__ uint32_t kv[XTILE + 40];
    __shared__ uint32_t sh[XW];
    const int r = blockIdx.x >> 1;
    const bool rev = blockIdx.x & 1;
    if (r >= n) return;
    uint32_t* cnt = rev ? cnt_rev : cnt_fwd;
    const uint32_t L = hlen[r];
    const int W = w - k + 1;
    // gates of PrepareSortedBatch (qualscore.cpp:65-66) + the reference's undefined n <= w-k case
    const int64_t nk = int64_t(L) - k;  // number of k-mers (the last one is never produced)
    const int64_t nw = nk - W + 1;      // number of windows
    if (status[r] != 0 || L < uint32_t(2 * k) || L < uint32_t(w) || nw <= 0 || (L + 15) / 16 > lds_words) {
        if (pass == 0 && threadIdx.x == 0) cnt[r] = 0;
        return;
    }
    const uint8_t* hs = hseq + offs[r];
    for (uint32_t wd = threadIdx.x; wd < (L + 15) / 16; wd += XB) {
        uint32_t v = 0;
        const uint32_t p0 = wd * 16;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (p0 + t < L) v |= uint32_t(hs[p0 + t] & 3u) << (2 * t);
        packed[wd] = v;
    }
    __syncthreads();
    const int kk = k > 16 ? 16 : k;      // bases that survive the 32-bit wrap
    const int skip = k - kk;
    const int64_t out_base = pass ? (rev ? off_rev[r] : off_fwd[r]) : 0;
    uint32_t emitted = 0;
    // tiles of XTILE windows; k-mer values of positions [t0 - 1, t0 + XTILE + W - 1) go to LDS
    for (int64_t t0 = 0; t0 < nw; t0 += XTILE) {
        for (int i = threadIdx.x; i < XTILE + W; i += XB) {
            const int64_t p = t0 - 1 + i;  // k-mer start
            uint32_t v = 0xFFFFFFFFu;
            if (p >= 0 && p < nk) {
                // The kk bases of the k-mer are 2 kk consecutive bits of the 2-bit image (at most two of its words): one
                // 64-bit funnel instead of a loop of kk LDS reads.  Forward strand: base t of the k-mer sits at bits
                // 2 t of the field and has to become digit kk-1-t — reverse the ORDER of the 2-bit groups (bit reversal,
                // then the two bits of every group swapped back).  Reverse strand: the k-mer at p reads the image
                // downwards from L-1-p and complements, so the field that ENDS at L-1-(p+skip) already has its first base
                // on top: complement it, no reversal.
                const uint32_t s0 = rev ? uint32_t(L - uint32_t(p + skip) - uint32_t(kk)) : uint32_t(p + skip);
                const uint32_t wd = s0 >> 4, sh = (s0 & 15u) * 2u;
                const unsigned long long two = (unsigned long long)packed[wd] | ((unsigned long long)(sh + 2u * uint32_t(kk) > 32u ? packed[wd + 1] : 0u) << 32);
                const uint32_t mask = kk == 16 ? 0xFFFFFFFFu : ((1u << (2 * kk)) - 1u);
                const uint32_t x = uint32_t(two >> sh) & mask;
                if (rev) {
                    v = ~x & mask;
                } else {
                    const uint32_t b = __brev(x);
                    v = (((b >> 1) & 0x55555555u) | ((b & 0x55555555u) << 1)) >> (32 - 2 * kk);
                }
            }
            kv[i] = v;
        }
        __syncthreads();
        const int64_t j = t0 + threadIdx.x;  // window index
        uint32_t flag = 0, amin = 0, vmin = 0;
        // leftmost argmin of the window that starts at kv index q
        auto argmin_at = [&](int q, uint32_t& best) {
            best = kv[q];
            int bi = 0;
            for (int t = 1; t < W; ++t) {
                const uint32_t v = kv[q + t];
                if (v < best) {
                    best = v;
                    bi = t;
                }
            }
            return bi;
        };
        int my_arg = -1;
        if (j < nw) {
            my_arg = argmin_at(int(threadIdx.x) + 1, vmin);
            amin = uint32_t(j + my_arg);
        }
        // the argmin of window j - 1 is the left neighbour's (lane shift inside a wave; the first lane of a wave, whose
        // neighbour sits in another wave or another tile, scans that window itself)
        {
            const int prev = __shfl_up(my_arg, 1);
            if (j < nw) {
                if (j == 0) {
                    flag = 1;
                } else {
                    int pi = prev;
                    if ((threadIdx.x & 63u) == 0) {
                        uint32_t pb;
                        pi = argmin_at(int(threadIdx.x), pb);
                    }
                    flag = (uint32_t(j - 1 + pi) != amin);
                }
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(flag, tot, sh);
        if (pass && flag) {
            omin[out_base + emitted + ex] = vmin;
            opos[out_base + emitted + ex] = amin;
        }
        emitted += tot;
        __syncthreads();
    }
    if (pass == 0 && threadIdx.x == 0) cnt[r] = emitted;
}

#define CKH(x)                           \
    do {                                 \
        hipError_t e_ = (x);             \
        if (e_ != hipSuccess) return e_; \
    } while (0)

extern "C" {

hipError_t iock_qual_scores(hipStream_t st, int n, const int64_t* offs, const uint8_t* qual, int k,
                            const double* tab_capped, const double* tab_nomin, double* score, double* err)
{
    if (n <= 0) return hipSuccess;
    const int waves_per_block = XB / 64;
    hipLaunchKernelGGL(k_qual_scores, dim3((n + waves_per_block - 1) / waves_per_block), dim3(XB), 0, st, n, offs, qual,
                       k, tab_capped, tab_nomin, score, err);
    return hipGetLastError();
}

hipError_t iock_hpc(hipStream_t st, int n, const int64_t* offs, const uint8_t* seq, const uint8_t* qual,
                    uint8_t* hseq, uint8_t* hqual, uint32_t* hlen, int32_t* status)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hpc, dim3(n), dim3(XB), 0, st, n, offs, seq, qual, hseq, hqual, hlen, status);
    return hipGetLastError();
}

hipError_t iock_hpc_error(hipStream_t st, int n, const int64_t* offs, const uint8_t* hqual, const uint32_t* hlen,
                          const double* tab_nomin, double* err)
{
    if (n <= 0) return hipSuccess;
    const int waves_per_block = XB / 64;
    hipLaunchKernelGGL(k_seq_error, dim3((n + waves_per_block - 1) / waves_per_block), dim3(XB), 0, st, n, offs, hqual,
                       hlen, tab_nomin, err);
    return hipGetLastError();
}

hipError_t iock_minimizers(hipStream_t st, int n, const int64_t* offs, const uint8_t* hseq, const uint32_t* hlen,
                           const int32_t* status, int k, int w, int pass, const int64_t* off_fwd,
                           const int64_t* off_rev, uint32_t* cnt_fwd, uint32_t* cnt_rev, uint32_t* omin,
                           uint32_t* opos, uint32_t max_hlen)
{
    if (n <= 0) return hipSuccess;
    uint32_t words = (max_hlen + 15) / 16 + 1;
    size_t lds = size_t(words) * 4;
    if (lds > 48 * 1024)
        CKH(hipFuncSetAttribute((const void*)k_minimizers, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_minimizers, dim3(2 * n), dim3(XB), lds, st, n, offs, hseq, hlen, status, k, w, pass, off_fwd,
                       off_rev, cnt_fwd, cnt_rev, omin, opos, words);
    return hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// C ABI: ioc_qual_scores / ioc_extract_minimizers / ioc_extracted_download / ioc_queries_from_extracted
// =====================================================================================================
namespace {

struct Tmp {
    void* p = nullptr;
    ~Tmp()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes)
    {
        const hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess) ioc_poison(p, bytes ? bytes : 16);
        return e;
    }
    template <class T>
    T* as()
    {
        return static_cast<T*>(p);
    }
};

struct Borrow {
    void* p;
    template <class T>
    T* as()
    {
        return static_cast<T*>(p);
    }
};

// InitQualTab / InitQualTabNomin, src/qualscore.cpp:156-180 (host libm pow, uploaded)
void qual_tables(double* capped, double* nomin)
{
    for (int i = 0; i < 129; ++i) capped[i] = nomin[i] = 0.0;
    for (int i = 33; i <= 128; ++i) {
        double v = pow(10, -((i - 33) / 10.0));
        nomin[i] = v;
        capped[i] = v > 0.79433 ? 0.79433 : v;
    }
}

int reserve_x(ioc_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return IOC_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return ioc_fail(c, IOC_ERR_CAPACITY, "hipMalloc failed in extraction");
    }
    b.cap = want;
    ioc_poison(b.p, want);
    return IOC_OK;
}

}  // namespace

#define XCHK(c, call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ioc_fail((c), IOC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int ioc_qual_scores(ioc_ctx* c, int32_t n, const int64_t* offs, const uint8_t* qual, int32_t k, double* score,
                    double* err_rate)
{
    if (!c || n < 0 || (n > 0 && (!offs || !qual || !score || !err_rate)) || k < 1) return IOC_ERR_ARG;
    XCHK(c, hipSetDevice(c->device));
    if (n == 0) return IOC_OK;
    const int64_t total = offs[n];
    Tmp d_offs, d_qual, d_tab, d_out;
    XCHK(c, d_offs.alloc(size_t(n + 1) * 8));
    XCHK(c, d_qual.alloc(size_t(total)));
    XCHK(c, d_tab.alloc(2 * 129 * 8));
    XCHK(c, d_out.alloc(size_t(n) * 16));
    double tabs[2 * 129];
    qual_tables(tabs, tabs + 129);
    hipStream_t s = c->stream;
    XCHK(c, hipMemcpyAsync(d_offs.p, offs, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(d_qual.p, qual, size_t(total), hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(d_tab.p, tabs, sizeof(tabs), hipMemcpyHostToDevice, s));
    XCHK(c, iock_qual_scores(s, n, d_offs.as<int64_t>(), d_qual.as<uint8_t>(), k, d_tab.as<double>(),
                             d_tab.as<double>() + 129, d_out.as<double>(), d_out.as<double>() + n));
    XCHK(c, hipMemcpyAsync(score, d_out.p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    XCHK(c, hipMemcpyAsync(err_rate, d_out.as<double>() + n, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    XCHK(c, hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i)
        if (std::isnan(score[i])) return ioc_fail(c, IOC_ERR_INPUT, "quality byte > 128 (the reference's table lookup throws)");
    return IOC_OK;
}

int ioc_extract_minimizers(ioc_ctx* c, int32_t n, const int64_t* offs, const uint8_t* seq, const uint8_t* qual,
                           int32_t k, int32_t w, uint32_t* hpc_len, double* hpc_err, int64_t* off_fwd,
                           int64_t* off_rev, int32_t* status)
{
    if (!c || n < 0 || (n > 0 && (!offs || !seq || !qual || !hpc_len || !hpc_err || !off_fwd || !off_rev || !status)))
        return IOC_ERR_ARG;
    if (k < 1 || k > 32 || w < k || w - k + 1 > 32) return ioc_fail(c, IOC_ERR_ARG, "need 1 <= k <= 32, k <= w <= k+31");
    XCHK(c, hipSetDevice(c->device));
    c->x_n = 0;
    c->x_total = 0;
    if (n == 0) return IOC_OK;
    const int64_t total = offs[n];
    hipStream_t s = c->stream;
    Tmp d_offs, d_seq, d_qual, d_status, d_err, d_cnt, d_tab;
    Borrow d_hseq{nullptr}, d_hqual{nullptr};  // the HPC strings stay in the context (ioc_extracted_hpc_download)
    XCHK(c, d_offs.alloc(size_t(n + 1) * 8));
    XCHK(c, d_seq.alloc(size_t(total)));
    XCHK(c, d_qual.alloc(size_t(total)));
    {
        int rc0;
        if ((rc0 = reserve_x(c, c->x_hseq, size_t(total))) != IOC_OK) return rc0;
        if ((rc0 = reserve_x(c, c->x_hqual, size_t(total))) != IOC_OK) return rc0;
        d_hseq.p = c->x_hseq.p;
        d_hqual.p = c->x_hqual.p;
    }
    XCHK(c, d_status.alloc(size_t(n) * 4));
    XCHK(c, d_err.alloc(size_t(n) * 8));
    XCHK(c, d_cnt.alloc(size_t(n) * 8));
    XCHK(c, d_tab.alloc(2 * 129 * 8));
    int rc;
    if ((rc = reserve_x(c, c->x_hpc_len, size_t(n) * 4)) != IOC_OK) return rc;
    if ((rc = reserve_x(c, c->x_off_fwd, size_t(n + 1) * 8)) != IOC_OK) return rc;
    if ((rc = reserve_x(c, c->x_off_rev, size_t(n + 1) * 8)) != IOC_OK) return rc;
    double tabs[2 * 129];
    qual_tables(tabs, tabs + 129);
    XCHK(c, hipMemcpyAsync(d_offs.p, offs, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(d_seq.p, seq, size_t(total), hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(d_qual.p, qual, size_t(total), hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(d_tab.p, tabs, sizeof(tabs), hipMemcpyHostToDevice, s));
    uint32_t* d_hlen = static_cast<uint32_t*>(c->x_hpc_len.p);
    XCHK(c, iock_hpc(s, n, d_offs.as<int64_t>(), d_seq.as<uint8_t>(), d_qual.as<uint8_t>(), d_hseq.as<uint8_t>(),
                     d_hqual.as<uint8_t>(), d_hlen, d_status.as<int32_t>()));
    XCHK(c, iock_hpc_error(s, n, d_offs.as<int64_t>(), d_hqual.as<uint8_t>(), d_hlen, d_tab.as<double>() + 129,
                           d_err.as<double>()));
    XCHK(c, hipMemcpyAsync(hpc_len, d_hlen, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    XCHK(c, hipMemcpyAsync(status, d_status.p, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    XCHK(c, hipMemcpyAsync(hpc_err, d_err.p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    XCHK(c, hipStreamSynchronize(s));
    uint32_t max_hlen = 1;
    for (int i = 0; i < n; ++i) {
        max_hlen = std::max(max_hlen, hpc_len[i]);
        // sort-stage gate (qualscore.cpp:65-73): HPC length < 2k or < w
        if (status[i] == 0 && (hpc_len[i] < uint32_t(2 * k) || hpc_len[i] < uint32_t(w))) status[i] = 1;
    }
    if ((uint64_t(max_hlen) + 15) / 16 * 4 > 64 * 1024 - 2048)
        return ioc_fail(c, IOC_ERR_CAPACITY, "a read has more than ~250k HPC bases");
    XCHK(c, hipMemcpyAsync(d_status.p, status, size_t(n) * 4, hipMemcpyHostToDevice, s));
    uint32_t* d_cf = d_cnt.as<uint32_t>();
    uint32_t* d_cr = d_cf + n;
    XCHK(c, iock_minimizers(s, n, d_offs.as<int64_t>(), d_hseq.as<uint8_t>(), d_hlen, d_status.as<int32_t>(), k, w, 0,
                            nullptr, nullptr, d_cf, d_cr, nullptr, nullptr, max_hlen));
    std::vector<uint32_t> cf, cr;
    cf.resize(size_t(n));
    cr.resize(size_t(n));
    XCHK(c, hipMemcpyAsync(cf.data(), d_cf, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    XCHK(c, hipMemcpyAsync(cr.data(), d_cr, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    XCHK(c, hipStreamSynchronize(s));
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        off_fwd[i] = tot;
        tot += cf[size_t(i)];
    }
    off_fwd[n] = tot;
    for (int i = 0; i < n; ++i) {
        off_rev[i] = tot;
        tot += cr[size_t(i)];
    }
    off_rev[n] = tot;
    if ((rc = reserve_x(c, c->x_min, size_t(tot) * 4)) != IOC_OK) return rc;
    if ((rc = reserve_x(c, c->x_pos, size_t(tot) * 4)) != IOC_OK) return rc;
    XCHK(c, hipMemcpyAsync(c->x_off_fwd.p, off_fwd, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
    XCHK(c, hipMemcpyAsync(c->x_off_rev.p, off_rev, size_t(n + 1) * 8, hipMemcpyHostToDevice, s));
    XCHK(c, iock_minimizers(s, n, d_offs.as<int64_t>(), d_hseq.as<uint8_t>(), d_hlen, d_status.as<int32_t>(), k, w, 1,
                            static_cast<int64_t*>(c->x_off_fwd.p), static_cast<int64_t*>(c->x_off_rev.p), d_cf, d_cr,
                            static_cast<uint32_t*>(c->x_min.p), static_cast<uint32_t*>(c->x_pos.p), max_hlen));
    XCHK(c, hipStreamSynchronize(s));
    c->x_n = n;
    c->x_total = tot;
    c->xh_off_fwd.assign(off_fwd, off_fwd + n + 1);
    c->xh_off_rev.assign(off_rev, off_rev + n + 1);
    c->xh_hpc_len.assign(hpc_len, hpc_len + n);
    c->xh_status.assign(status, status + n);
    c->xh_offs.assign(offs, offs + n + 1);
    return IOC_OK;
}

int ioc_extracted_hpc_download(ioc_ctx* c, char* hpc_seq, char* hpc_qual, int64_t cap)
{
    if (!c || !hpc_seq || !hpc_qual) return IOC_ERR_ARG;
    XCHK(c, hipSetDevice(c->device));
    const int n = c->x_n;
    if (n <= 0) return ioc_fail(c, IOC_ERR_STATE, "ioc_extract_minimizers first");
    const int64_t total = c->xh_offs[size_t(n)];
    if (cap < total) return ioc_fail(c, IOC_ERR_CAPACITY, "buffer smaller than the raw sequence bytes");
    // layout: read i's HPC string occupies [offs[i], offs[i] + hpc_len[i]) (capacity = raw length)
    XCHK(c, hipMemcpyAsync(hpc_seq, c->x_hseq.p, size_t(total), hipMemcpyDeviceToHost, c->stream));
    XCHK(c, hipMemcpyAsync(hpc_qual, c->x_hqual.p, size_t(total), hipMemcpyDeviceToHost, c->stream));
    XCHK(c, hipStreamSynchronize(c->stream));
    static const char L[5] = {'A', 'C', 'G', 'T', 'N'};
    for (int i = 0; i < n; ++i)
        for (int64_t t = c->xh_offs[size_t(i)]; t < c->xh_offs[size_t(i)] + c->xh_hpc_len[size_t(i)]; ++t)
            hpc_seq[t] = L[uint8_t(hpc_seq[t]) > 3 ? 4 : uint8_t(hpc_seq[t])];
    return IOC_OK;
}

int ioc_extracted_download(ioc_ctx* c, uint32_t* min_val, uint32_t* min_pos, int64_t cap)
{
    if (!c || !min_val || !min_pos) return IOC_ERR_ARG;
    XCHK(c, hipSetDevice(c->device));
    if (cap < c->x_total) return ioc_fail(c, IOC_ERR_CAPACITY, "buffer smaller than the extracted minimizers");
    if (c->x_total == 0) return IOC_OK;
    XCHK(c, hipMemcpyAsync(min_val, c->x_min.p, size_t(c->x_total) * 4, hipMemcpyDeviceToHost, c->stream));
    XCHK(c, hipMemcpyAsync(min_pos, c->x_pos.p, size_t(c->x_total) * 4, hipMemcpyDeviceToHost, c->stream));
    XCHK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_queries_from_extracted(ioc_ctx* c, const uint8_t* keep, const uint8_t* err_cell, const uint32_t* min_total)
{
    if (!c || !keep || !err_cell || !min_total) return IOC_ERR_ARG;
    XCHK(c, hipSetDevice(c->device));
    const int n = c->x_n;
    if (n <= 0) return ioc_fail(c, IOC_ERR_STATE, "ioc_extract_minimizers first");
    std::vector<uint8_t> cell(err_cell, err_cell + n);
    for (int i = 0; i < n; ++i)
        if (!keep[i] || cell[size_t(i)] < 1 || cell[size_t(i)] > 15) cell[size_t(i)] = 1;
    int rc;
    if ((rc = reserve_x(c, c->b_err_cell, size_t(n))) != IOC_OK) return rc;
    if ((rc = reserve_x(c, c->b_min_total, size_t(n) * 4)) != IOC_OK) return rc;
    XCHK(c, hipMemcpyAsync(c->b_err_cell.p, cell.data(), size_t(n), hipMemcpyHostToDevice, c->stream));
    XCHK(c, hipMemcpyAsync(c->b_min_total.p, min_total, size_t(n) * 4, hipMemcpyHostToDevice, c->stream));
    XCHK(c, hipStreamSynchronize(c->stream));
    rc = ioc_queries_bind_device(c, n, static_cast<int64_t*>(c->x_off_fwd.p), static_cast<int64_t*>(c->x_off_rev.p),
                                 static_cast<uint32_t*>(c->x_min.p), static_cast<uint32_t*>(c->x_pos.p), c->x_total,
                                 static_cast<uint32_t*>(c->x_hpc_len.p), static_cast<uint8_t*>(c->b_err_cell.p),
                                 static_cast<uint32_t*>(c->b_min_total.p), c->xh_off_fwd.data(), c->xh_off_rev.data());
    if (rc != IOC_OK) return rc;
    // entries the caller gated out never become clusters (cluster.cpp:116-160)
    for (int i = 0; i < n; ++i)
        if (!keep[i]) c->h_forced_t[size_t(i)] = -2;
    c->forced_dirty = true;
    c->x_keep.assign(keep, keep + n);
    return IOC_OK;
}

}  // extern "C"
