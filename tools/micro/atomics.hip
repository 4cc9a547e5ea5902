// Microbenchmark behind the index-build design: 12 M random global atomics, as k_hash_insert_queries issues them
// (returning atomicAdd on a 2 MB counter array) against the same number of non-returning atomicOr into a 66 MB bitmap and of
// non-returning atomicAdd into the counter array.   hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_ret_add(uint32_t* cnt, uint32_t mask, uint32_t* out, uint32_t per)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per; ++i) acc += atomicAdd(&cnt[mix(t * per + i) & mask], 1u);
    out[t] = acc;
}
__global__ void k_noret_add(uint32_t* cnt, uint32_t mask, uint32_t per)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) atomicAdd(&cnt[mix(t * per + i) & mask], 1u);
}
__global__ void k_noret_or(uint32_t* bm, uint32_t words, uint32_t per)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t h = mix(t * per + i);
        atomicOr(&bm[h % words], 1u << (h >> 27));
    }
}
__global__ void k_plain_store(uint32_t* bm, uint32_t words, uint32_t per)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t h = mix(t * per + i);
        bm[h % words] = h;
    }
}
int main()
{
    const uint32_t threads = 3000 * 256, per = 16;  // 12.3 M operations
    uint32_t *cnt, *bm, *out;
    const uint32_t slots = 1u << 19, words = 66u << 18;  // 2 MB of counters, 66 MB of bitmap
    hipMalloc(&cnt, slots * 4); hipMalloc(&bm, size_t(words) * 4); hipMalloc(&out, threads * 4);
    hipMemset(cnt, 0, slots * 4); hipMemset(bm, 0, size_t(words) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto&& launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
        printf("%-44s %8.3f ms  %6.1f G ops/s\n", name, best, double(threads) * per / best * 1e-6);
    };
    run("returning atomicAdd, 2 MB counters", [&] { hipLaunchKernelGGL(k_ret_add, dim3(3000), dim3(256), 0, 0, cnt, slots - 1, out, per); });
    run("non-returning atomicAdd, 2 MB counters", [&] { hipLaunchKernelGGL(k_noret_add, dim3(3000), dim3(256), 0, 0, cnt, slots - 1, per); });
    run("non-returning atomicOr, 66 MB bitmap", [&] { hipLaunchKernelGGL(k_noret_or, dim3(3000), dim3(256), 0, 0, bm, words, per); });
    run("plain store, 66 MB", [&] { hipLaunchKernelGGL(k_plain_store, dim3(3000), dim3(256), 0, 0, bm, words, per); });
    return 0;
}
