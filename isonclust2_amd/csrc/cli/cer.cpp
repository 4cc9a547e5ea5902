// cer.cpp — binary (.cer) reader / writer for the batch record (see cer.hpp for the layout rules).
#include "cer.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <sys/uio.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unordered_map>

namespace cer {
namespace {

// The writer gathers: small fields are staged in blocks that never move, large ones (bases, qualities, minimizer arrays — mostly
// views of the input archive's mapping) are handed to the kernel where they lie, IOV_MAX pieces per writev.  460 MB in ~45 ms on the
// MI355X hosts' page cache, against 90 – 220 ms through stdio from heap copies (tools/micro/file_write.cpp, profiles/r05_cli_breakdown.txt).
struct Writer {
    int fd;
    bool ok = true;
    uint32_t next_ptr = 1;
    static constexpr size_t BLK = size_t(1) << 20, DIRECT = 2048, MAXIOV = 1024;
    std::vector<iovec> iov;
    std::vector<std::unique_ptr<char[]>> blocks;
    ptrdiff_t cur = -1;  // current staging block (none yet) and its fill
    size_t used = 0;
    explicit Writer(int fd_) : fd(fd_) { iov.reserve(MAXIOV); }
    void raw(const void* p, size_t n)
    {
        if (!n) return;
        if (n >= DIRECT) {
            iov.push_back(iovec{const_cast<void*>(p), n});
        } else {
            if (cur < 0 || used + n > BLK) {
                if (size_t(++cur) >= blocks.size()) blocks.emplace_back(new char[BLK]);
                used = 0;
            }
            char* d = blocks[size_t(cur)].get() + used;
            memcpy(d, p, n);
            used += n;
            if (!iov.empty() && static_cast<char*>(iov.back().iov_base) + iov.back().iov_len == d)
                iov.back().iov_len += n;
            else
                iov.push_back(iovec{d, n});
        }
        if (iov.size() >= MAXIOV) flush();
    }
    void flush()
    {
        size_t i = 0;
        while (ok && i < iov.size()) {
            const ssize_t w = writev(fd, iov.data() + i, int(std::min(iov.size() - i, MAXIOV)));
            if (w < 0) {
                if (errno == EINTR) continue;
                ok = false;
                break;
            }
            size_t left = size_t(w);
            while (i < iov.size() && left >= iov[i].iov_len) left -= iov[i++].iov_len;
            if (left) {
                iov[i].iov_base = static_cast<char*>(iov[i].iov_base) + left;
                iov[i].iov_len -= left;
            }
        }
        iov.clear();
        cur = -1;  // (what the blocks held is in the file)
    }
    template <class T>
    void pod(T v) { raw(&v, sizeof(T)); }
    void str(const std::string& s)
    {
        pod<uint64_t>(s.size());
        raw(s.data(), s.size());
    }
    void str(const Bytes& s)
    {
        pod<uint64_t>(s.size());
        raw(s.data(), s.size());
    }
    uint32_t new_shared() { return (next_ptr++) | 0x80000000u; }
};

// the archive's read-only mapping; the views of a loaded batch share it
struct Mapping {
    int fd = -1;
    void* p = MAP_FAILED;
    size_t n = 0;
    Mapping() = default;
    Mapping(const Mapping&) = delete;
    Mapping& operator=(const Mapping&) = delete;
    ~Mapping()
    {
        if (p != MAP_FAILED) munmap(p, n);
        if (fd >= 0) close(fd);
    }
};

struct Reader {
    const uint8_t* p;
    const uint8_t* e;
    std::shared_ptr<const void> keep;
    bool ok = true;
    void raw(void* d, size_t n)
    {
        if (size_t(e - p) < n) {
            ok = false;
            memset(d, 0, n);
            return;
        }
        memcpy(d, p, n);
        p += n;
    }
    template <class T>
    T pod()
    {
        T v;
        raw(&v, sizeof(T));
        return v;
    }
    std::string str()
    {
        uint64_t n = pod<uint64_t>();
        if (!ok || uint64_t(e - p) < n) {
            ok = false;
            return std::string();
        }
        std::string s(reinterpret_cast<const char*>(p), size_t(n));
        p += n;
        return s;
    }
    Bytes bytes()  // a view: nothing is copied
    {
        uint64_t n = pod<uint64_t>();
        if (!ok || uint64_t(e - p) < n) {
            ok = false;
            return Bytes();
        }
        Bytes b(reinterpret_cast<const char*>(p), size_t(n), keep);
        p += n;
        return b;
    }
    template <class T>
    Span<T> span()
    {
        uint64_t n = pod<uint64_t>();
        if (!ok || n > uint64_t(e - p) / sizeof(T)) {  // (no multiplication: a crafted count must not wrap)
            ok = false;
            return Span<T>();
        }
        Span<T> v(reinterpret_cast<const T*>(p), size_t(n), keep);
        p += size_t(n) * sizeof(T);
        return v;
    }
};

void put_seq(Writer& w, const Seq& s)
{
    w.str(s.name);
    w.str(s.seq);
    w.str(s.qual);
    w.pod<double>(s.score);
    w.pod<double>(s.errorRate);
}
void get_seq(Reader& r, Seq& s)
{
    s.name = r.str();
    s.seq = r.bytes();
    s.qual = r.bytes();
    s.score = r.pod<double>();
    s.errorRate = r.pod<double>();
}
void put_useq(Writer& w, const std::unique_ptr<Seq>& s)
{
    w.pod<uint8_t>(s ? 1 : 0);
    if (s) put_seq(w, *s);
}
void get_useq(Reader& r, std::unique_ptr<Seq>& s)
{
    if (r.pod<uint8_t>()) {
        s.reset(new Seq);
        get_seq(r, *s);
    } else {
        s.reset();
    }
}
void put_mins(Writer& w, const Span<Minimizer>& m)
{
    w.pod<uint64_t>(m.size());
    if (!m.empty()) w.raw(m.data(), m.size() * sizeof(Minimizer));
}
void get_mins(Reader& r, Span<Minimizer>& m) { m = r.span<Minimizer>(); }
void put_args(Writer& w, const CmdArgs& a)
{
    w.pod<uint8_t>(a.Verbose);
    w.pod<uint8_t>(a.Debug);
    w.str(a.InFastq);
    for (int32_t v : {a.KmerSize, a.BatchSize, a.BatchMaxSeq, a.WindowSize, a.MinShared, a.ConsMinSize, a.ConsMaxSize,
                      a.ConsPeriod, a.MinClsSize})
        w.pod<int32_t>(v);
    for (double v : {a.MinQual, a.MappedThreshold, a.AlignedThreshold, a.MinFraction, a.MinProbNoHits}) w.pod<double>(v);
    w.str(a.BatchOutFolder);
    w.pod<int32_t>(a.Mode);
}
void get_args(Reader& r, CmdArgs& a)
{
    a.Verbose = r.pod<uint8_t>() != 0;
    a.Debug = r.pod<uint8_t>() != 0;
    a.InFastq = r.str();
    for (int32_t* v : {&a.KmerSize, &a.BatchSize, &a.BatchMaxSeq, &a.WindowSize, &a.MinShared, &a.ConsMinSize,
                       &a.ConsMaxSize, &a.ConsPeriod, &a.MinClsSize})
        *v = r.pod<int32_t>();
    for (double* v : {&a.MinQual, &a.MappedThreshold, &a.AlignedThreshold, &a.MinFraction, &a.MinProbNoHits})
        *v = r.pod<double>();
    a.BatchOutFolder = r.str();
    a.Mode = r.pod<int32_t>();
}

}  // namespace

bool save_batch(const Batch& b, const std::string& path, std::string& err)
{
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        err = "Failed to open " + path + "!";
        return false;
    }
    Writer w(fd);
    w.pod<int32_t>(b.BatchNr);
    w.pod<uint64_t>(b.BatchStart);
    w.pod<uint64_t>(b.BatchEnd);
    w.pod<uint64_t>(b.BatchBases);
    w.pod<int32_t>(b.TotalReads);
    w.pod<int32_t>(b.NrCls);
    put_args(w, b.SortArgs);
    w.str(b.LeftLeaf);
    w.str(b.RightLeaf);
    w.pod<int32_t>(b.Depth);
    w.pod<uint64_t>(b.Db.size());
    for (auto& kv : b.Db) {
        w.pod<uint32_t>(kv.first);
        w.pod<uint64_t>(kv.second.size());
        if (!kv.second.empty()) w.raw(kv.second.data(), kv.second.size() * 4);
    }
    w.pod<uint64_t>(b.Cls.size());
    for (auto& c : b.Cls) {
        if (!c) {
            w.pod<uint32_t>(0);
            continue;
        }
        w.pod<uint32_t>(w.new_shared());
        w.pod<uint64_t>(c->size());
        for (auto& ps : *c) {
            if (!ps) {
                w.pod<uint32_t>(0);
                continue;
            }
            w.pod<uint32_t>(w.new_shared());
            put_useq(w, ps->RawSeq);
            put_useq(w, ps->HpcSeq);
            put_mins(w, ps->Mins);
            put_mins(w, ps->RevMins);
            w.pod<int32_t>(ps->MatchStrand);
            w.str(ps->Id);
        }
    }
    w.pod<uint64_t>(b.NrConsGs);
    for (uint64_t i = 0; i < b.NrConsGs; ++i) {
        if (i < b.ConsGs.size() && !b.ConsGs[i].empty()) {
            w.pod<uint8_t>(1);
            w.pod<uint64_t>(b.ConsGs[i].size());
            w.raw(b.ConsGs[i].data(), b.ConsGs[i].size());
        } else {
            w.pod<uint8_t>(0);  // null unique_ptr<spoa::Graph>
        }
    }
    w.flush();
    bool ok = close(fd) == 0 && w.ok;
    if (!ok) err = "Failed to write " + path + "!";
    return ok;
}

bool load_batch(Batch& b, const std::string& path, std::string& err)
{
    // the archive stays mapped read-only for as long as a view of it lives; only the small fields are copied out
    auto mh = std::make_shared<Mapping>();
    Mapping& mp = *mh;
    mp.fd = open(path.c_str(), O_RDONLY);
    if (mp.fd < 0) {
        err = "Failed to load batch " + path + ": cannot open";
        return false;
    }
    struct stat sb;
    if (fstat(mp.fd, &sb) != 0) {
        err = "Failed to load batch " + path + ": cannot stat";
        return false;
    }
    mp.n = size_t(sb.st_size);
    static const uint8_t none = 0;
    const uint8_t* base = &none;
    if (mp.n > 0) {
        mp.p = mmap(nullptr, mp.n, PROT_READ, MAP_PRIVATE, mp.fd, 0);
        if (mp.p == MAP_FAILED) {
            err = "Failed to load batch " + path + ": short read";
            return false;
        }
        base = static_cast<const uint8_t*>(mp.p);
        // page tables for the whole file now, on a few threads (2 – 4 ms for 460 MB in the page cache) instead of one fault per
        // 64 KB while the records are walked; MADV_POPULATE_READ is Linux 5.14: where it is refused the faults come as they come
        if (mp.n >= (size_t(32) << 20)) {
            const size_t nt = 4, step = ((mp.n / nt) + 4095) & ~size_t(4095);
            char* const at = static_cast<char*>(mp.p);
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; ++t) {
                const size_t a0 = std::min(mp.n, t * step), a1 = std::min(mp.n, (t + 1) * step);
                if (a1 > a0) th.emplace_back([at, a0, a1] { (void)madvise(at + a0, a1 - a0, 22 /* MADV_POPULATE_READ */); });
            }
            for (auto& x : th) x.join();
        }
    }
    Reader r{base, base + mp.n, mh};
    b = Batch();
    b.BatchNr = r.pod<int32_t>();
    b.BatchStart = r.pod<uint64_t>();
    b.BatchEnd = r.pod<uint64_t>();
    b.BatchBases = r.pod<uint64_t>();
    b.TotalReads = r.pod<int32_t>();
    b.NrCls = r.pod<int32_t>();
    get_args(r, b.SortArgs);
    b.LeftLeaf = r.str();
    b.RightLeaf = r.str();
    b.Depth = r.pod<int32_t>();
    uint64_t nk = r.pod<uint64_t>();
    for (uint64_t i = 0; r.ok && i < nk; ++i) {
        uint32_t key = r.pod<uint32_t>();
        uint64_t m = r.pod<uint64_t>();
        if (m > uint64_t(r.e - r.p) / 4) {  // (compared without multiplying: a crafted 64-bit count must not wrap)
            r.ok = false;
            break;
        }
        b.Db.emplace_back(key, Span<uint32_t>(reinterpret_cast<const uint32_t*>(r.p), size_t(m), mh));
        r.p += size_t(m) * 4;
    }
    if (!std::is_sorted(b.Db.begin(), b.Db.end(), [](const MinDB::value_type& x, const MinDB::value_type& y) { return x.first < y.first; }))
        std::sort(b.Db.begin(), b.Db.end(), [](const MinDB::value_type& x, const MinDB::value_type& y) { return x.first < y.first; });
    std::unordered_map<uint32_t, std::shared_ptr<Cluster>> seenC;
    std::unordered_map<uint32_t, std::shared_ptr<ProcSeq>> seenP;
    uint64_t nc = r.pod<uint64_t>();
    for (uint64_t i = 0; r.ok && i < nc; ++i) {
        uint32_t id = r.pod<uint32_t>();
        if (id == 0) {
            b.Cls.push_back(nullptr);
            continue;
        }
        if (!(id & 0x80000000u)) {
            b.Cls.push_back(seenC[id]);
            continue;
        }
        auto c = std::make_shared<Cluster>();
        seenC[id & 0x7FFFFFFFu] = c;
        uint64_t m = r.pod<uint64_t>();
        for (uint64_t j = 0; r.ok && j < m; ++j) {
            uint32_t pid = r.pod<uint32_t>();
            if (pid == 0) {
                c->push_back(nullptr);
                continue;
            }
            if (!(pid & 0x80000000u)) {
                c->push_back(seenP[pid]);
                continue;
            }
            auto ps = std::make_shared<ProcSeq>();
            seenP[pid & 0x7FFFFFFFu] = ps;
            get_useq(r, ps->RawSeq);
            get_useq(r, ps->HpcSeq);
            get_mins(r, ps->Mins);
            get_mins(r, ps->RevMins);
            ps->MatchStrand = r.pod<int32_t>();
            ps->Id = r.str();
            c->push_back(ps);
        }
        b.Cls.push_back(c);
    }
    b.NrConsGs = r.pod<uint64_t>();
    b.ConsGs.clear();
    for (uint64_t i = 0; r.ok && i < b.NrConsGs; ++i) {
        b.ConsGs.emplace_back();
        if (r.pod<uint8_t>() == 0) continue;
        const uint64_t n = r.pod<uint64_t>();
        if (!r.ok || uint64_t(r.e - r.p) < n) {
            r.ok = false;
            break;
        }
        // a non-null graph must be this build's blob: the reference writes spoa's own members here, without a length
        // prefix (spoa's cereal layout is not in the reference tree) — such a file cannot be exchanged in consensus mode
        if (n >= 8 && memcmp(r.p, "IOCPOA1", 8) == 0) {
            err = "Failed to load batch " + path + ": consensus graph " + std::to_string(i) +
                  " was written by an earlier build (IOCPOA1: its edge weights and node order are not this build's; a consensus "
                  "grown on it would be neither build's) — cluster the batch again with this build";
            return false;
        }
        if (n < 8 || memcmp(r.p, "IOCPOA2", 8) != 0) {
            err = "Failed to load batch " + path + ": consensus graph " + std::to_string(i) +
                  " is not in this build's format (spoa-serialized graphs of the reference are not readable here; "
                  "files written with consensus off exchange fine)";
            return false;
        }
        b.ConsGs.back().assign(r.p, r.p + n);
        r.p += n;
    }
    if (!r.ok) {
        err = "Failed to load batch " + path + ": truncated or corrupt archive";
        return false;
    }
    return true;
}

bool map_file(const std::string& path, MappedFile& out, std::string& err)
{
    auto mh = std::make_shared<Mapping>();
    mh->fd = open(path.c_str(), O_RDONLY);
    if (mh->fd < 0) {
        err = "Failed to open " + path + "!";
        return false;
    }
    struct stat sb;
    if (fstat(mh->fd, &sb) != 0) {
        err = "Failed to stat " + path + "!";
        return false;
    }
    mh->n = size_t(sb.st_size);
    out = MappedFile();
    if (mh->n == 0) return true;
    mh->p = mmap(nullptr, mh->n, PROT_READ, MAP_PRIVATE, mh->fd, 0);
    if (mh->p == MAP_FAILED) {
        err = "Failed to map " + path + "!";
        return false;
    }
    if (mh->n >= (size_t(32) << 20)) {  // (page tables now, on a few threads: see load_batch)
        const size_t nt = 4, step = ((mh->n / nt) + 4095) & ~size_t(4095);
        char* const at = static_cast<char*>(mh->p);
        const size_t total = mh->n;
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t) {
            const size_t a0 = std::min(total, t * step), a1 = std::min(total, (t + 1) * step);
            if (a1 > a0) th.emplace_back([at, a0, a1] { (void)madvise(at + a0, a1 - a0, 22 /* MADV_POPULATE_READ */); });
        }
        for (auto& x : th) x.join();
    }
    out.data = static_cast<const char*>(mh->p);
    out.size = mh->n;
    out.keep = mh;
    return true;
}

struct GatherFile::Impl {
    int fd = -1;
    Writer* w = nullptr;
};
GatherFile::GatherFile() : p_(new Impl) {}
GatherFile::~GatherFile()
{
    (void)close();
    delete p_;
}
bool GatherFile::open(const std::string& path)
{
    (void)close();
    p_->fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (p_->fd < 0) return false;
    p_->w = new Writer(p_->fd);
    return true;
}
void GatherFile::put(const void* data, size_t n)
{
    if (p_->w) p_->w->raw(data, n);
}
bool GatherFile::close()
{
    if (!p_->w) return true;
    p_->w->flush();
    const bool ok = ::close(p_->fd) == 0 && p_->w->ok;
    delete p_->w;
    p_->w = nullptr;
    p_->fd = -1;
    return ok;
}

std::shared_ptr<void> huge_alloc(size_t bytes)
{
    const size_t n = (std::max<size_t>(bytes, 1) + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    void* p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    (void)madvise(p, n, MADV_HUGEPAGE);
    return std::shared_ptr<void>(p, [n](void* q) { munmap(q, n); });
}

bool save_sorted_idx(const std::string& fastq_path, const std::string& path)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    uint64_t n = fastq_path.size();
    bool ok = fwrite(&n, 8, 1, f) == 1 && fwrite(fastq_path.data(), 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

bool load_sorted_idx(std::string& fastq_path, const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint64_t n = 0;
    bool ok = fread(&n, 8, 1, f) == 1 && n < (1u << 20);
    if (ok) {
        fastq_path.resize(size_t(n));
        ok = fread(&fastq_path[0], 1, size_t(n), f) == n;
    }
    fclose(f);
    return ok;
}

}  // namespace cer
