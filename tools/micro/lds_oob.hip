// What does gfx950 do with an LDS atomic whose address lies beyond the workgroup's LDS allocation?  (Developer probe: the
// scoring kernel could drop its per-posting window test if such an atomic is discarded by the hardware.)
//  1. boundary scan: one workgroup with D bytes of dynamic LDS writes a marker to every dword address in [D - 256, D + 8192)
//     and reads it back: the first address that does not hold its marker is where the allocation ends for the hardware;
//  2. isolation: many workgroups per CU fill their allocation with a pattern, then every lane issues ds_add_u32 to addresses
//     from the hardware end upwards (up to 512 KB above), then every workgroup checks its pattern: a changed word means an
//     out-of-range atomic landed in somebody's memory.
//   hipcc --offload-arch=gfx950 -O3 -o lds_oob tools/micro/lds_oob.hip && ./lds_oob
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void ds_store(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ds_load(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void ds_add(uint32_t addr, uint32_t v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

__global__ void __launch_bounds__(64) k_scan(uint32_t dyn, uint32_t* out, uint32_t span)
{
    extern __shared__ uint32_t lds[];
    const uint32_t base = uint32_t(reinterpret_cast<uintptr_t>(lds));
    for (uint32_t a = threadIdx.x * 4; a < span; a += 64 * 4) {
        const uint32_t addr = base + dyn - 256 + a;
        ds_store(addr, 0xA5000000u | a);
        out[a / 4] = ds_load(addr) == (0xA5000000u | a) ? 1u : 0u;
    }
    if (threadIdx.x == 0) out[span / 4] = base;
}

__global__ void __launch_bounds__(256) k_iso(uint32_t words, uint32_t hw_end, uint32_t* bad, int rounds)
{
    extern __shared__ uint32_t lds[];
    const uint32_t base = uint32_t(reinterpret_cast<uintptr_t>(lds));
    for (uint32_t i = threadIdx.x; i < words; i += 256) lds[i] = 0x5A000000u ^ (i * 2654435761u) ^ blockIdx.x;
    __syncthreads();
    uint32_t x = blockIdx.x * 977u + threadIdx.x * 31u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const uint32_t off = (x >> 8) % (512u * 1024u / 4u);  // dword offsets above the end, up to 512 KB
        ds_add(base + hw_end + off * 4u, 1u);
        if ((r & 7) == 0) ds_add(base + (x % words) * 4u, 0u);  // (and in-range atomics that change nothing, mixed in)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_sleep(100);
    __syncthreads();
    uint32_t nb = 0;
    for (uint32_t i = threadIdx.x; i < words; i += 256)
        if (lds[i] != (0x5A000000u ^ (i * 2654435761u) ^ blockIdx.x)) ++nb;
    if (nb) atomicAdd(bad, nb);
}

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("%s: %s\n", #x, hipGetErrorString(e));                       \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main()
{
    const uint32_t span = 256 + 8192;
    uint32_t* d_out;
    CK(hipMalloc(&d_out, (span / 4 + 1) * 4));
    std::vector<uint32_t> h(span / 4 + 1);
    uint32_t hw_end_of[4] = {0, 0, 0, 0};
    const uint32_t dyns[4] = {1024, 5000, 24576, 40000};
    for (int t = 0; t < 4; ++t) {
        const uint32_t dyn = dyns[t];
        CK(hipFuncSetAttribute((const void*)k_scan, hipFuncAttributeMaxDynamicSharedMemorySize, int(dyn)));
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), dyn, 0, dyn, d_out, span);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
        uint32_t first_bad = span;
        for (uint32_t a = 0; a < span; a += 4)
            if (!h[a / 4]) {
                first_bad = a;
                break;
            }
        uint32_t later_ok = 0;
        for (uint32_t a = first_bad; a < span; a += 4) later_ok += h[a / 4];
        const uint32_t hw_end = dyn - 256 + first_bad;  // relative to the dynamic base
        hw_end_of[t] = hw_end;
        printf("dynamic LDS %6u B at base %u: stores hold up to offset %u (= %u above the request; 1280-granule end would be %u); %u addresses above that hold a store\n",
               dyn, h[span / 4], hw_end, hw_end - dyn, (h[span / 4] + dyn + 1279) / 1280 * 1280 - h[span / 4], later_ok);
    }
    uint32_t* d_bad;
    CK(hipMalloc(&d_bad, 4));
    for (int t = 0; t < 4; ++t) {
        const uint32_t dyn = dyns[t];
        CK(hipMemset(d_bad, 0, 4));
        CK(hipFuncSetAttribute((const void*)k_iso, hipFuncAttributeMaxDynamicSharedMemorySize, int(dyn)));
        hipLaunchKernelGGL(k_iso, dim3(8192), dim3(256), dyn, 0, dyn / 4, hw_end_of[t], d_bad, 4096);
        CK(hipDeviceSynchronize());
        uint32_t bad = 0;
        CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("isolation, %6u B per workgroup, 8192 workgroups x 256 lanes x 4096 atomics above the end: %u words changed\n", dyn, bad);
    }
    return 0;
}
