// Issue rate of packed 16-bit integer VALU ops against 32-bit ones on gfx950 (developer microbenchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 17 + i;
    uint32_t b = seed * 3 + 1, c = seed ^ 0x1234;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {  // 32-bit: add, max, max
                int x = int(a[i]) + int(b);
                x = max(x, int(c));
                a[i] = uint32_t(max(x, int(a[(i + 1) & 15])));
            } else if (MODE == 1) {  // packed: pk_add, pk_max, pk_max
                uint32_t x;
                asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(x) : "v"(a[i]), "v"(b));
                asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
                asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(a[(i + 1) & 15]));
            } else if (MODE == 2) {  // perm + pk_sub clamp + pk_max
                uint32_t x;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x) : "v"(a[i]), "v"(b), "v"(c));
                asm volatile("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(x) : "v"(x), "v"(c));
                asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(a[(i + 1) & 15]));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name)
{
    uint32_t* d;
    hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = 4096.0 * 256 * iters * 16 * 3;
    printf("%s: %.2f ms, %.2f T lane-instr/s\n", name, ms, ops / ms / 1e9);
    hipFree(d);
}
int main()
{
    run<0>("i32 add/max/max");
    run<1>("pk16 add/max/max");
    run<2>("perm/pk_sub clamp/pk_max");
    // saturation semantics of the clamp bit
    return 0;
}
