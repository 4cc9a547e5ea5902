// How long does the host wait after a tiny kernel + 12-byte read-back into pinned memory: hipStreamSynchronize against polling the
// pinned words?  (The resolve ends every sweep with such a read-back: DESIGN 5.4.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
__global__ void k_set(uint32_t* p, uint32_t v) { p[0] = v; p[1] = v + 1; p[2] = v + 2; }
__global__ void k_spin(uint32_t* p, int iters) { uint32_t x = p[0]; for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u; if (x == 42) p[3] = x; }
int main()
{
    uint32_t* d; volatile uint32_t* h;
    hipMalloc(&d, 64); hipHostMalloc((void**)&h, 64, hipHostMallocDefault);
    hipStream_t s; hipStreamCreate(&s);
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int work : {0, 20000}) {   // (kernel of ~0 and ~50 us before the read-back)
        for (int mode = 0; mode < 2; ++mode) {
            double tot = 0; const int R = 200;
            for (int r = 0; r < R + 10; ++r) {
                h[2] = 0xFFFFFFFEu;
                const double t0 = now();
                if (work) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, work);
                hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s, d, uint32_t(r));
                hipMemcpyAsync((void*)h, d, 12, hipMemcpyDeviceToHost, s);
                if (mode == 0) hipStreamSynchronize(s);
                else while (h[2] == 0xFFFFFFFEu) { }
                const double t1 = now();
                if (r >= 10) tot += t1 - t0;
                if (mode == 1) hipStreamSynchronize(s);
                if (h[0] != uint32_t(r) || h[2] != uint32_t(r) + 2) { printf("bad read-back\n"); return 1; }
            }
            printf("kernel work %5d: %-22s %.1f us per round trip (launch + kernel + 12-byte read-back + wait)\n", work, mode ? "polling pinned memory" : "hipStreamSynchronize", tot / R);
        }
    }
    return 0;
}
