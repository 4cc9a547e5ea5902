// What a one-shot GPU process costs outside its work: runtime start, VRAM / host memory of a given size,
// and — with the harness tools/micro/proc_cost.py — the time before main and after _exit.
// proc_cost VRAM_MB HOST_MB [free] [reset] [exit]     prints one JSON line with CLOCK_MONOTONIC stamps
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(uint32_t* p, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = uint32_t(i);
}
int main(int argc, char** argv)
{
    const double t0 = now_ms();
    const size_t vram = size_t(argc > 1 ? atoll(argv[1]) : 0) << 20, host = size_t(argc > 2 ? atoll(argv[2]) : 0) << 20;
    bool do_free = false, do_reset = false, clean_exit = false;
    for (int i = 3; i < argc; ++i) {
        do_free |= !strcmp(argv[i], "free");
        do_reset |= !strcmp(argv[i], "reset");
        clean_exit |= !strcmp(argv[i], "exit");
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 1;
    const double t1 = now_ms();
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
    const double t2 = now_ms();
    uint32_t* d = nullptr;
    if (vram) {
        if (hipMalloc(&d, vram) != hipSuccess) return 2;
    }
    const double t3 = now_ms();
    if (vram) {
        k_touch<<<4096, 256, 0, s>>>(d, vram / 4);
        if (hipStreamSynchronize(s) != hipSuccess) return 3;
    }
    const double t4 = now_ms();
    char* h = nullptr;
    if (host) {
        h = static_cast<char*>(malloc(host));
        memset(h, 1, host);
    }
    const double t5 = now_ms();
    if (do_free && d) (void)hipFree(d);
    if (do_free && h) free(h);
    if (do_reset) (void)hipDeviceReset();
    const double t6 = now_ms();
    printf("{\"t_begin_mono_ms\": %.3f, \"init\": %.1f, \"stream\": %.1f, \"malloc\": %.1f, \"touch\": %.1f, \"host\": %.1f, \"free_reset\": %.1f, \"t_end_mono_ms\": %.3f}\n", t0,
           t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, now_ms());
    fflush(stdout);
    if (clean_exit) return 0;
    _exit(0);
}
