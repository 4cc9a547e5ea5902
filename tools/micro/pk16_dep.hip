// Latency of DEPENDENT VALU ops on gfx950: a single chain per wave, one wave per SIMD (developer microbenchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x1234;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (MODE == 0) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (MODE == 2) { asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)); asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(c) : "v"(c), "v"(b)); }
            if (MODE == 3) { asm volatile("v_max_i32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)); asm volatile("v_max_i32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(b)); }
            if (MODE == 4) asm volatile("v_pk_max_i16 %0, %1, %2\n s_nop 0" : "=v"(a) : "v"(a), "v"(b));
            if (MODE == 5) asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (MODE == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ c;
}
template <int MODE>
void run(const char* name, int ops_per)
{
    uint32_t* d;
    hipMalloc(&d, 1024 * 64 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  -> %.2f ns per op (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / (double(iters) * 64 * ops_per), ms * 1e6 / (double(iters) * 64 * ops_per) * 2.4);
    hipFree(d);
}
int main()
{
    run<0>("dependent v_max_i32", 1);
    run<1>("dependent v_pk_max_i16", 1);
    run<2>("two chains v_pk_max_i16", 2);
    run<3>("two chains v_max_i32", 2);
    run<4>("dependent v_pk_max_i16 + s_nop 0", 1);
    run<5>("dependent v_pk_sub_i16", 1);
    run<6>("dependent v_perm_b32", 1);
    return 0;
}
