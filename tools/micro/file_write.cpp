// How fast can a process put ~460 MB (config 2's output .cer in fast mode) into a NEW file, from a source that is itself a mapped file?
// file_write DIR [MB]   — one line per method.  g++ -O2 -pthread -o file_write.bin file_write.cpp
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t n = size_t(argc > 2 ? atoll(argv[2]) : 460) << 20;
    const std::string src = dir + "/fw_src.bin";
    {   // the source file (page cache hot)
        int fd = open(src.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        std::vector<char> b(1 << 20);
        for (size_t i = 0; i < b.size(); ++i) b[i] = char(i * 131);
        for (size_t o = 0; o < n; o += b.size()) (void)!write(fd, b.data(), b.size());
        close(fd);
    }
    int sfd = open(src.c_str(), O_RDONLY);
    double t0 = now();
    const char* m = static_cast<const char*>(mmap(nullptr, n, PROT_READ, MAP_PRIVATE, sfd, 0));
    printf("mmap source: %.1f ms\n", now() - t0);
    auto par = [&](int nt, auto fn) {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(t, nt); });
        for (auto& x : th) x.join();
    };
    for (int nt : {1, 4, 16}) {
        t0 = now();
        par(nt, [&](int t, int k) {
            const size_t a = n / k * t, b = t == k - 1 ? n : n / k * (t + 1);
            (void)madvise(const_cast<char*>(m) + a, b - a, MADV_POPULATE_READ);
        });
        printf("MADV_POPULATE_READ of the source, %d threads: %.1f ms\n", nt, now() - t0);
        if (nt == 1) {
            munmap(const_cast<char*>(m), n);
            m = static_cast<const char*>(mmap(nullptr, n, PROT_READ, MAP_PRIVATE, sfd, 0));
        }
    }
    auto fresh = [&](const char* name) {
        const std::string p = dir + "/" + name;
        unlink(p.c_str());
        return open(p.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
    };
    for (size_t chunk : {size_t(64) << 10, size_t(1) << 20, size_t(16) << 20}) {
        int fd = fresh("fw_a.bin");
        t0 = now();
        for (size_t o = 0; o < n; o += chunk) (void)!write(fd, m + o, std::min(chunk, n - o));
        double t1 = now();
        close(fd);
        printf("write() from the mapping, %zu KB chunks: %.1f ms (%.2f GB/s), close %.1f\n", chunk >> 10, t1 - t0, n / (t1 - t0) / 1e6, now() - t1);
    }
    {
        int fd = fresh("fw_v.bin");
        t0 = now();
        std::vector<iovec> iv;
        for (size_t o = 0; o < n; o += 48 << 10) {
            iv.push_back(iovec{const_cast<char*>(m) + o, std::min<size_t>(48 << 10, n - o)});
            if (iv.size() == 1024 || o + (48 << 10) >= n) {
                (void)!writev(fd, iv.data(), int(iv.size()));
                iv.clear();
            }
        }
        printf("writev, 48 KB pieces x 1024: %.1f ms\n", now() - t0);
        close(fd);
    }
    {
        std::vector<char> heap(n);
        memcpy(heap.data(), m, n);
        int fd = fresh("fw_h.bin");
        t0 = now();
        for (size_t o = 0; o < n; o += 1 << 20) (void)!write(fd, heap.data() + o, std::min<size_t>(1 << 20, n - o));
        printf("write() from the heap, 1 MB chunks: %.1f ms\n", now() - t0);
        close(fd);
    }
    for (int nt : {2, 4, 8, 16}) {
        int fd = fresh("fw_p.bin");
        t0 = now();
        (void)!ftruncate(fd, off_t(n));
        par(nt, [&](int t, int k) {
            const size_t a = n / k * t, b = t == k - 1 ? n : n / k * (t + 1);
            for (size_t o = a; o < b; o += 1 << 20) (void)!pwrite(fd, m + o, std::min<size_t>(1 << 20, b - o), off_t(o));
        });
        printf("pwrite, %d threads on disjoint ranges: %.1f ms\n", nt, now() - t0);
        close(fd);
    }
    for (int nt : {1, 4, 16}) {
        int fd = fresh("fw_m.bin");
        t0 = now();
        (void)!ftruncate(fd, off_t(n));
        char* w = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        par(nt, [&](int t, int k) {
            const size_t a = n / k * t, b = t == k - 1 ? n : n / k * (t + 1);
            memcpy(w + a, m + a, b - a);
        });
        double t1 = now();
        munmap(w, n);
        close(fd);
        printf("MAP_SHARED + memcpy, %d threads: %.1f ms (+ munmap/close %.1f)\n", nt, t1 - t0, now() - t1);
    }
    {
        int fd = fresh("fw_c.bin");
        t0 = now();
        off_t oi = 0;
        size_t left = n;
        while (left) {
            ssize_t r = copy_file_range(sfd, &oi, fd, nullptr, left, 0);
            if (r <= 0) { printf("copy_file_range failed\n"); break; }
            left -= size_t(r);
        }
        printf("copy_file_range, whole file: %.1f ms\n", now() - t0);
        close(fd);
    }
    {
        int fd = fresh("fw_c2.bin");
        t0 = now();
        for (size_t o = 0; o < n; o += 150000) {   // record-sized pieces at odd offsets
            off_t oi = off_t(o) + 13;
            size_t left = std::min<size_t>(150000, n - o - 13);
            while (left) {
                ssize_t r = copy_file_range(sfd, &oi, fd, nullptr, left, 0);
                if (r <= 0) break;
                left -= size_t(r);
            }
        }
        printf("copy_file_range, 150 kB pieces at odd offsets: %.1f ms\n", now() - t0);
        close(fd);
    }
    {   // what anonymous memory of that size costs: plain, and with MADV_HUGEPAGE
        for (int huge = 0; huge < 2; ++huge) {
            t0 = now();
            char* a = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
            if (huge) (void)madvise(a, n, MADV_HUGEPAGE);
            memset(a, 1, n);
            double t1 = now();
            munmap(a, n);
            printf("anonymous %zu MB %s: touch %.1f ms, unmap %.1f ms\n", n >> 20, huge ? "MADV_HUGEPAGE" : "4K pages", t1 - t0, now() - t1);
        }
    }
    for (const char* f : {"fw_src.bin", "fw_a.bin", "fw_v.bin", "fw_h.bin", "fw_p.bin", "fw_m.bin", "fw_c.bin", "fw_c2.bin"}) unlink((dir + "/" + f).c_str());
    return 0;
}
