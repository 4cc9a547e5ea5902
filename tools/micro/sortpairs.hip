// How fast does rocPRIM sort the 12 M (minimizer value, query) pairs of an index build (22 value bits)?  Context for DESIGN 5.2.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
int main()
{
    const size_t n = 12300000;
    std::vector<uint32_t> hk(n);
    std::vector<uint16_t> hv(n);
    uint32_t x = 12345;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; hk[i] = (x >> 8) % 3075822u; hv[i] = uint16_t(i / 4100); }
    uint32_t *k0, *k1; uint16_t *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 2); hipMalloc(&v1, n * 2);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), n * 2, hipMemcpyHostToDevice);
    size_t tmp = 0; void* d_tmp = nullptr;
    for (int bits : {22, 32}) {
        rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 0, bits);
        hipMalloc(&d_tmp, tmp);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(a);
            rocprim::radix_sort_pairs(d_tmp, tmp, k0, k1, v0, v1, n, 0, bits);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (r) best = ms < best ? ms : best;
        }
        printf("rocprim::radix_sort_pairs, %zu pairs (u32 key, %d bits; u16 value): %.3f ms (temp %zu MB)\n", n, bits, best, tmp >> 20);
        hipFree(d_tmp);
    }
    return 0;
}
