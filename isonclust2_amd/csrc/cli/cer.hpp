// cer.hpp — the batch record and its binary (.cer) image.
//
// Field order and encoding follow the reference's serialize() members (src/serialize.h:38-43,
// src/args.h:32-35, src/cluster_data.h:24, src/seq.h:63, src/minimizer.h:27) under the conventions of
// cereal's portable-less BinaryOutputArchive (native little endian; arithmetic = raw bytes, bool 1 B,
// enum = int32, std::string / std::vector = u64 count + payload, unordered_map = u64 count + (key,
// value) pairs, unique_ptr = u8 valid flag, shared_ptr = u32 id with the MSB set on first occurrence).
// cereal itself is absent from the reference tree (vendor/cereal is an empty submodule), so the byte
// layout is UNVERIFIED against a reference-written file: it is self-consistent between this tool's
// sort / cluster / dump / info.  spoa graphs (ConsGs) are written as null entries, or — consensus mode — as the
// graph blobs of this build's POA engine.
#ifndef IOC_CER_HPP
#define IOC_CER_HPP

#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

namespace cer {

// (packed: a view of the archive's mapping starts at whatever offset the record has in the file, so every access must be
// compiled as an unaligned one)
struct __attribute__((packed)) Minimizer {
    uint32_t Min, Pos, Index;
};

// The large fields of a record — bases, qualities, minimizer arrays: 150 kB per read, 460 MB per 3000-read batch — are IMMUTABLE
// once written, so the in-memory record holds them as views: of the archive's read-only mapping (load_batch copies nothing; the
// view keeps the mapping alive), or of a buffer of its own (records made by `sort`, consensus representatives).  A copy of a
// record shares the bytes.  What a one-shot `cluster` process gains (profiles/r05_cli_breakdown.txt): no 460 MB of page faults on
// load, none of the 150 kB deep copies per new representative, a writer that hands the kernel the mapped bytes (writev), and a
// process exit that does not have to give 1.5 GB of anonymous memory back page by page.
template <class T>
class Span {
    const T* p_ = nullptr;
    size_t n_ = 0;
    std::shared_ptr<const void> keep_;

public:
    Span() = default;
    Span(const T* p, size_t n, std::shared_ptr<const void> keep) : p_(p), n_(n), keep_(std::move(keep)) {}
    Span(std::vector<T>&& v) { *this = std::move(v); }
    Span(const std::vector<T>& v) { *this = std::vector<T>(v); }
    Span(std::initializer_list<T> l) { *this = std::vector<T>(l); }
    Span& operator=(std::vector<T>&& v)
    {
        auto own = std::make_shared<std::vector<T>>(std::move(v));
        p_ = own->data();
        n_ = own->size();
        keep_ = std::move(own);
        return *this;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    const T* data() const { return p_; }
    T operator[](size_t i) const  // (by value through memcpy: a view of a mapped file is not aligned for T)
    {
        T v;
        memcpy(&v, reinterpret_cast<const char*>(p_) + i * sizeof(T), sizeof(T));
        return v;
    }
    void append_to(std::vector<T>& dst) const
    {
        const size_t at = dst.size();
        dst.resize(at + n_);
        if (n_) memcpy(static_cast<void*>(dst.data() + at), p_, n_ * sizeof(T));
    }
    bool operator==(const std::vector<T>& o) const { return n_ == o.size() && (n_ == 0 || memcmp(p_, o.data(), n_ * sizeof(T)) == 0); }
    void clear()
    {
        p_ = nullptr;
        n_ = 0;
        keep_.reset();
    }
};

class Bytes {
    const char* p_ = "";
    size_t n_ = 0;
    std::shared_ptr<const void> keep_;

public:
    Bytes() = default;
    Bytes(const char* p, size_t n, std::shared_ptr<const void> keep) : p_(p), n_(n), keep_(std::move(keep)) {}
    Bytes(std::string&& s) { *this = std::move(s); }
    Bytes(const std::string& s) { *this = std::string(s); }
    Bytes(const char* z) { *this = std::string(z); }
    Bytes& operator=(std::string&& s)
    {
        auto own = std::make_shared<std::string>(std::move(s));
        p_ = own->data();
        n_ = own->size();
        keep_ = std::move(own);
        return *this;
    }
    Bytes& assign(const char* p, size_t n) { return *this = std::string(p, n); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    const char* data() const { return p_; }
    const char* begin() const { return p_; }
    const char* end() const { return p_ + n_; }
    char operator[](size_t i) const { return p_[i]; }
    std::string str() const { return std::string(p_, n_); }
    bool operator==(const Bytes& o) const { return n_ == o.n_ && (n_ == 0 || memcmp(p_, o.p_, n_) == 0); }
    bool operator==(const std::string& o) const { return n_ == o.size() && (n_ == 0 || memcmp(p_, o.data(), n_) == 0); }
    bool operator==(const char* z) const { return n_ == strlen(z) && (n_ == 0 || memcmp(p_, z, n_) == 0); }
    template <class U>
    bool operator!=(const U& o) const { return !(*this == o); }
    friend std::ostream& operator<<(std::ostream& os, const Bytes& b) { return os.write(b.p_, std::streamsize(b.n_)); }
};

struct Seq {  // src/seq.h:20-98
    std::string name;
    Bytes seq, qual;
    double score = 0, errorRate = 0;
};

struct ProcSeq {  // src/cluster_data.h:14-26
    std::unique_ptr<Seq> RawSeq, HpcSeq;
    Span<Minimizer> Mins, RevMins;
    int32_t MatchStrand = 0;
    std::string Id;
};
typedef std::vector<std::shared_ptr<ProcSeq>> Cluster;
typedef std::vector<std::shared_ptr<Cluster>> Clusters;

enum ClsMode : int32_t { Sahlin = 0, Fast = 1, Furious = 2, None = 3 };  // src/args.h:7

struct CmdArgs {  // src/args.h:9-37
    bool Verbose = false, Debug = false;
    std::string InFastq;
    int32_t KmerSize = 11, BatchSize = 50000, BatchMaxSeq = 30000, WindowSize = 15, MinShared = 5;
    int32_t ConsMinSize = 50, ConsMaxSize = -150, ConsPeriod = 500, MinClsSize = 3;
    double MinQual = 7.0, MappedThreshold = 0.65, AlignedThreshold = 0.2, MinFraction = 0.8, MinProbNoHits = 0.1;
    std::string BatchOutFolder = "isONclust2_batches";
    int32_t Mode = Sahlin;
};

typedef std::vector<std::pair<uint32_t, Span<uint32_t>>> MinDB;  // kept sorted by key in memory; posting lists are views (of the mapping, of one flat export)

struct Batch {  // src/serialize.h:23-43
    int32_t BatchNr = 0;
    uint64_t BatchStart = 0, BatchEnd = 0, BatchBases = 0;
    int32_t TotalReads = 0, NrCls = 0;
    CmdArgs SortArgs;
    std::string LeftLeaf, RightLeaf;
    int32_t Depth = 0;
    MinDB Db;
    Clusters Cls;
    uint64_t NrConsGs = 0;
    // ConsGs (src/serialize.h:21,37): one graph per cluster.  spoa's own cereal layout is not in the tree; a graph is
    // written as unique_ptr flag + u64 length + the blob of ioc_poa_graph_save (empty vector entry = null pointer).
    std::vector<std::vector<uint8_t>> ConsGs;
};

// anonymous memory for the large flat host arrays, in transparent huge pages where the system grants them on madvise (a page
// fault per 2 MB instead of per 4 KB: 29 against 90 ms for 460 MB on the MI355X hosts, tools/micro/file_write.cpp); never
// shrinks, freed with the holder
std::shared_ptr<void> huge_alloc(size_t bytes);

// a file mapped read-only as a whole (what load_batch reads from; `sort` parses its FASTQ out of one: the reads' bases and
// qualities are views of it); nullptr + err when it cannot be had.  size 0: an empty file (data() is then a valid empty string).
struct MappedFile {
    const char* data = "";
    size_t size = 0;
    std::shared_ptr<const void> keep;  // what the views of it hold
};
bool map_file(const std::string& path, MappedFile& out, std::string& err);

// a file written as a gather of pieces that stay where they are until flush() / close() (writev, 1024 pieces a call): small pieces
// may be passed with copy = true (staged), large ones — views of a mapping — are handed to the kernel in place
class GatherFile {
    struct Impl;
    Impl* p_;

public:
    GatherFile();
    ~GatherFile();
    GatherFile(const GatherFile&) = delete;
    GatherFile& operator=(const GatherFile&) = delete;
    bool open(const std::string& path);
    void put(const void* data, size_t n);  // (pieces below 2 KB are staged: the caller's copy may go)
    bool close();                          // false: some write failed
};

bool save_batch(const Batch& b, const std::string& path, std::string& err);
bool load_batch(Batch& b, const std::string& path, std::string& err);
bool save_sorted_idx(const std::string& fastq_path, const std::string& path);  // SortedIdx, src/output.h:15-23
bool load_sorted_idx(std::string& fastq_path, const std::string& path);

}  // namespace cer
#endif
