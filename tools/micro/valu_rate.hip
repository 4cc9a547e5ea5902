// Issue rate of the integer / float VALU instructions the aligner's forward pass is made of, on gfx950, at 1 / 2 / 4 / 8
// waves per SIMD (developer microbenchmark; its output is kept under profiles/ and sets bench.py's VALU peak).
// Every wave runs 16 independent dependency chains of the instruction under test; a workgroup is 256 threads = one wave
// per SIMD, an LDS reservation caps the workgroups per CU at the wanted number and the grid is exactly CUs x that number,
// so every SIMD holds exactly `wps` waves for the whole launch.  Launches run >= 50 ms after a burn-in (the clock under an
// all-VALU load settles well below the 2.4 GHz nominal); the shader clock during the launch is read as
// s_memtime ticks per s_memrealtime tick (100 MHz) by one wave per workgroup.
// Round 4: every wave works for a fixed stretch of WALL time (the 100 MHz clock, read every 256 x 16 instructions) and counts what it
// got done, instead of a fixed number of iterations: with equal work per wave the waves of a SIMD do not finish together — the
// issue arbiter serves the oldest wave first, the first workgroups of a launch left after a third of it and the last one ran
// alone at the end (round 3's "workgroups alive 52 - 86 % of the launch") — so the tail of every launch ran at the rate of fewer
// waves than the row claims.  Now all `wps` waves of every SIMD are at work for the whole measured stretch.
// Round 3: the launches are 50 - 100 ms (were 1 - 5: the launch tail was a third of the timed region), the rate and the
// "cycles of a SIMD per wave-instruction" both come from the EVENT time of the launch (the per-wave cycle counts of round 2
// under-counted when waves did not all start together: the sub-2-cycle v_add_u32 reading), and the line says how much of
// the launch the average workgroup was alive.  Added: the instruction mix of the two-pairs-per-wave forward pass.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

enum { FMA_F32, ADD_U32, MAX_I32, MAX3_I32, ADD_SDWA, PK_MAX_I16, PK_ADD_I16, MOV_DPP, MIX_ALIGN, PK_MAX_U16, PERM_B32, MIX_V2, N_MODES };
static const char* NAMES[N_MODES] = {"v_fma_f32", "v_add_u32", "v_max_i32", "v_max3_i32", "v_add_u32_sdwa (byte sel)",
                                     "v_pk_max_i16", "v_pk_add_i16", "v_mov_b32_dpp wave_shr:1",
                                     "aligner cell mix (add_sdwa, max3, sub, max, max)", "v_pk_max_u16", "v_perm_b32",
                                     "aligner v2 cell-pair mix (perm, add, 4 pk_max_u16, sub)"};
static const int OPS[N_MODES] = {1, 1, 1, 1, 1, 1, 1, 1, 5, 1, 1, 7};

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint64_t* cyc, int iters, uint32_t seed, uint64_t budget)
{
    extern __shared__ uint32_t pad[];
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 17 + i;
    uint32_t b = seed * 3 + 1, c = seed ^ 0x1234;
    float fb = 1.0001f, fc = 0.5f;
    const uint64_t t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    uint64_t chunks = 0;
    // (budget == 0: `iters` iterations, the burn-in; otherwise chunks of 256 iterations until the wall-clock budget is spent)
    for (;;) {
    if (budget ? wall_clock64() - w0 >= budget : chunks > 0) break;
    const int n_it = budget ? 256 : iters;
    ++chunks;
    for (int it = 0; it < n_it; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));
            if (MODE == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == MAX3_I32) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b));
            if (MODE == PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == PK_ADD_I16) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == MIX_V2) {
                uint32_t inc, t, fn, e, hh;
                asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(fn) : "v"(a[(i + 1) & 15]), "v"(c));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(inc) : "v"(b), "v"(c), "v"(b));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a[(i + 2) & 15]), "v"(inc));
                asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(fn));
                asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(e) : "v"(a[i]), "v"(b));
                asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(hh) : "v"(t), "v"(e));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(hh), "v"(c));
            }
            if (MODE == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (MODE == MIX_ALIGN) {
                uint32_t x, e, f;
                asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(x) : "v"(a[i]), "v"(b));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(e) : "v"(a[(i + 1) & 15]), "v"(c));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(f) : "v"(a[(i + 2) & 15]), "v"(b));
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(e), "v"(f));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(c));
            }
        }
    }
    }
    const uint64_t t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[0] * 0;
    if ((threadIdx.x & 63) == 0) {  // per wave: cycles, wall ticks, chunks done, start
        uint64_t* o = cyc + 4 * (size_t(blockIdx.x) * 4 + (threadIdx.x >> 6));
        o[0] = t1 - t0;
        o[1] = w1 - w0;
        o[2] = chunks;
        o[3] = w0;
    }
}

template <int MODE>
void run(int n_cu, int wps, double* out_rate)
{
    const int grid = n_cu * wps;
    const uint64_t budget = 4000000;  // 40 ms of the 100 MHz clock
    uint32_t* d;
    uint64_t* dc;
    hipMalloc(&d, size_t(grid) * 256 * 4);
    hipMalloc(&dc, size_t(grid) * 4 * 32);
    // at most `wps` workgroups per CU: each takes just over 160 KB / (wps + 1) of LDS (64 KB is the default cap per block)
    size_t lds = wps >= 8 ? 0 : (size_t(160) * 1024 / (wps + 1) + 1024) & ~size_t(255);
    if (lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    }
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, dc, 200000 / wps / OPS[MODE], 1u, uint64_t(0));  // burn-in
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, dc, 0, 1u, budget);
    hipDeviceSynchronize();
    const int nw = grid * 4;
    uint64_t* hc = new uint64_t[4 * size_t(nw)];
    hipMemcpy(hc, dc, size_t(nw) * 32, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0, chunks = 0;
    uint64_t first_start = ~0ull, last_start = 0, cmin = ~0ull, cmax = 0;
    for (int i = 0; i < nw; ++i) {
        cyc += double(hc[4 * i]);
        ticks += double(hc[4 * i + 1]);
        chunks += double(hc[4 * i + 2]);
        cmin = hc[4 * i + 2] < cmin ? hc[4 * i + 2] : cmin;
        cmax = hc[4 * i + 2] > cmax ? hc[4 * i + 2] : cmax;
        first_start = hc[4 * i + 3] < first_start ? hc[4 * i + 3] : first_start;
        last_start = hc[4 * i + 3] > last_start ? hc[4 * i + 3] : last_start;
    }
    delete[] hc;
    const double ghz = cyc / ticks * 0.1;            // s_memtime ticks per 10 ns
    const double secs = ticks / nw * 1e-8;           // the stretch a wave was at work (the budget + its last chunk)
    const double winstr = chunks * 256.0 * 16 * OPS[MODE];      // wave-instructions of the launch
    const double rate = winstr * 64 / secs / 1e12;
    // cycles of a SIMD per wave-instruction it issued = SIMD-cycles of the stretch / instructions of the SIMD's waves
    const double cpi = secs * ghz * 1e9 * double(n_cu) * 4 / winstr;
    printf("  %-56s wps %d: %6.2f T lane-op/s  %5.2f cycles of a SIMD per wave-instr  (clock %.2f GHz; %.1f ms per wave, all started within %.3f ms; slowest / fastest wave got %.2f of the mean work: %.2f / %.2f)\n",
           NAMES[MODE], wps, rate, cpi, ghz, secs * 1e3, double(last_start - first_start) * 1e-5, double(cmin) / (chunks / nw), double(cmin) / (chunks / nw), double(cmax) / (chunks / nw));
    if (out_rate) *out_rate = rate;
    hipFree(d);
    hipFree(dc);
}

template <int MODE>
void sweep(int n_cu, double* best)
{
    for (int wps : {1, 2, 3, 4, 8}) {
        double r = 0;
        run<MODE>(n_cu, wps, &r);
        if (r > best[MODE]) best[MODE] = r;
    }
}

int main()
{
    int n_cu = 256, clk = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device: %d CUs, nominal clock %.2f GHz; SIMDs %d\n", n_cu, clk / 1e6, n_cu * 4);
    double best[N_MODES] = {0};
    sweep<FMA_F32>(n_cu, best);
    sweep<ADD_U32>(n_cu, best);
    sweep<MAX_I32>(n_cu, best);
    sweep<MAX3_I32>(n_cu, best);
    sweep<ADD_SDWA>(n_cu, best);
    sweep<PK_MAX_I16>(n_cu, best);
    sweep<PK_ADD_I16>(n_cu, best);
    sweep<MOV_DPP>(n_cu, best);
    sweep<MIX_ALIGN>(n_cu, best);
    sweep<PK_MAX_U16>(n_cu, best);
    sweep<PERM_B32>(n_cu, best);
    sweep<MIX_V2>(n_cu, best);
    printf("JSON {");
    for (int m = 0; m < N_MODES; ++m) printf("%s\"%s\": %.3f", m ? ", " : "", NAMES[m], best[m]);
    printf("}\n");
    return 0;
}
