// isONclust2-hip — the `isONclust2 sort | cluster | dump | info` command line on top of the MI355X path.
//
// Mirrors the sub-commands and flags of the reference (src/main.cpp:29-73, src/args.cpp) so that the
// external batch-and-merge pipeline (README.md:105-117) can call it unchanged; every hot-path
// computation goes through the C ABI of libisonclust2_hip.so (no CPU fallback: without a GPU the
// tool exits with an error).  Host code here is I/O and bookkeeping only: FASTQ parsing, the batching
// policy of `sort` (main.cpp:149-199), the member moves of `cluster` (cluster.cpp:177-261), the TSV /
// FASTQ writers of `dump` (output.cpp:151-275).
#include <dlfcn.h>
#include <fcntl.h>
#include <getopt.h>
#include <poll.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <numeric>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "cer.hpp"
#include "isonclust2_hip.h"

using namespace cer;
using std::cerr;
using std::endl;
using std::string;

static const char* VERSION = "2.4-hip-r1";
static bool VERBOSE = false;

// A served job (see "serve" below) must not take the resident process down with it: its error exits unwind to the server loop.
struct JobExit {
    int rc;
};
static bool g_served = false;
static ioc_ctx* g_srv_ctx = nullptr;  // the resident process's context, made by its first job

[[noreturn]] static void die(const string& m)
{
    cerr << m << endl;
    if (g_served) throw JobExit{1};
    // (no unwinding: `sort` calls this from a batch-writer thread while others still run, and the reference's own error exits
    // leave through exit(1) with nothing left to flush)
    fflush(nullptr);
    _exit(1);
}

static string table_path()
{
    if (const char* e = getenv("ISONCLUST2_TABLE")) return e;
    char buf[4096];
    ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    string exe = n > 0 ? string(buf, size_t(n)) : string(".");
    size_t p = exe.rfind('/');
    string dir = p == string::npos ? "." : exe.substr(0, p);
    return dir + "/../data/pmin_shared.bin";
}

static ioc_ctx* make_ctx()
{
    ioc_ctx* c = nullptr;
    int dev = 0;
    if (const char* e = getenv("ISONCLUST2_DEVICE")) dev = atoi(e);
    int rc = ioc_ctx_create(dev, &c);
    if (rc != IOC_OK) die("No usable MI355X device (ioc_ctx_create failed with " + std::to_string(rc) + "); there is no CPU fallback.");
    return c;
}

static void check(ioc_ctx* c, int rc, const char* what)
{
    if (rc < 0) die(string(what) + ": " + ioc_last_error(c));
}

static int dir_exists(const string& p)
{
    struct stat info;
    return stat(p.c_str(), &info) == 0 && (info.st_mode & S_IFDIR);
}
static void create_outdir(const string& d)
{
    if (dir_exists(d)) {
        cerr << "Warning: reusing existing output directory: " << d << endl;
        return;
    }
    if (mkdir(d.c_str(), 0755) != 0) die("Failed to create output directory!");
}
static void create_file(const string& p, std::ofstream& f)
{
    f.open(p);
    if (!f.is_open()) die("Failed to open " + p + "!");
}

static void print_batch_info(const Batch& b)
{
    int ncls = 0, nnt = 0;
    for (auto& c : b.Cls)
        if (c && !c->empty() && c->at(0) && c->at(0)->RawSeq && c->at(0)->RawSeq->score > -1) {
            ncls++;
            if (c->size() > 2) nnt++;
        }
    cerr << "\tBatch number: " << b.BatchNr << endl;
    cerr << "\tBatch range: [" << b.BatchStart << "," << b.BatchEnd << "]" << endl;
    cerr << "\tDepth: " << b.Depth << endl;
    cerr << "\tNr sequences: " << b.BatchEnd - b.BatchStart + 1 << endl;
    cerr << "\tNr bases: " << b.BatchBases << endl;
    cerr << "\tNr clusters: " << ncls << endl;
    cerr << "\tNr nontrivial clusters: " << nnt << endl;
    cerr << "\tMinimizers in database: " << b.Db.size() << endl;
    size_t ng = 0;
    for (auto& g : b.ConsGs) ng += !g.empty();
    if (ng)  // (beyond the reference's lines: the one field of a .cer that is not exchangeable with the reference)
        cerr << "\tConsensus graphs: " << ng << " in this build's own format (the reference serializes spoa graphs; not interchangeable)" << endl;
}

static int parse_mode(const string& m)
{
    if (m == "sahlin") return Sahlin;
    if (m == "fast") return Fast;
    if (m == "furious") return Furious;
    die("Illegal clustering mode: " + m);
}

// ===================================================================================================
// sort  (src/main.cpp:75-202)
// ===================================================================================================
static int main_sort(int argc, char** argv)
{
    static const struct option lo[] = {
        {"version", no_argument, 0, 'V'}, {"verbose", no_argument, 0, 'v'}, {"debug", no_argument, 0, 'd'},
        {"mode", required_argument, 0, 'x'}, {"help", no_argument, 0, 'h'}, {"kmer-size", required_argument, 0, 'k'},
        {"window-size", required_argument, 0, 'w'}, {"min-shared", required_argument, 0, 'm'},
        {"mapped-threshold", required_argument, 0, 'r'}, {"aligned-threshold", required_argument, 0, 'a'},
        {"min-fraction", required_argument, 0, 'f'}, {"min-prob-no-hits", required_argument, 0, 'p'},
        {"min-qual", required_argument, 0, 'q'}, {"min-cls-size", required_argument, 0, 'F'},
        {"low-cons-size", required_argument, 0, 'g'}, {"max-cons-size", required_argument, 0, 'c'},
        {"cons-period", required_argument, 0, 'P'}, {"outfolder", required_argument, 0, 'o'},
        {"batch-size", required_argument, 0, 'B'}, {"batch-max-seq", required_argument, 0, 'M'}, {0, 0, 0, 0}};
    CmdArgs a;
    int o;
    while ((o = getopt_long(argc, argv, "k:w:dhvo:m:r:a:f:p:q:B:x:g:c:M:P:F:", lo, nullptr)) != -1) {
        switch (o) {
            case 'k': a.KmerSize = atoi(optarg); break;
            case 'w': a.WindowSize = atoi(optarg); break;
            case 'm': a.MinShared = atoi(optarg); break;
            case 'r': a.MappedThreshold = atof(optarg); break;
            case 'a': a.AlignedThreshold = atof(optarg); break;
            case 'f': a.MinFraction = atof(optarg); break;
            case 'p': a.MinProbNoHits = atof(optarg); break;
            case 'q': a.MinQual = atof(optarg); break;
            case 'B': a.BatchSize = atoi(optarg); break;
            case 'M': a.BatchMaxSeq = atoi(optarg); break;
            case 'P': a.ConsPeriod = atoi(optarg); break;
            case 'g': a.ConsMinSize = atoi(optarg); break;
            case 'c': a.ConsMaxSize = atoi(optarg); break;
            case 'F': a.MinClsSize = atoi(optarg); break;
            case 'o': a.BatchOutFolder = optarg; break;
            case 'v': a.Verbose = true; break;
            case 'd': a.Debug = true; break;
            case 'x': a.Mode = parse_mode(optarg); break;
            case 'h': cerr << "isONclust2-hip sort [options] reads.fastq  (flags as `isONclust2 sort`)" << endl; exit(0);
            default: break;
        }
    }
    // src/args.cpp:135-148
    if (a.KmerSize < 10 || a.KmerSize > 31) die("Invalid kmer size (must be in [10,31])!");
    if (a.KmerSize > a.WindowSize) die("The window size must be larger than or equal to the kmer size!");
    if (optind >= argc) die("No input fastq specified!");
    a.InFastq = argv[optind];
    VERBOSE = a.Verbose;

    const bool trace = getenv("IOC_TRACE") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ioc] sort: %-44s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_lap).count());
        t_lap = t1;
    };
    // ---- FASTQ (whole file in RAM, like bioparser Parse(-1), main.cpp:109-112) ----
    auto ctx_future = std::async(std::launch::async, [] { return make_ctx(); });  // (the runtime starts beside the parsing)
    // (parsed out of a read-only mapping: a read's bases and qualities are views of it — nothing of an 8 GB file is copied)
    MappedFile fqmap;
    {
        string merr;
        if (!map_file(a.InFastq, fqmap, merr)) die("Failed to open " + a.InFastq + "!");
    }
    std::vector<Seq> reads;
    {
        const char* p = fqmap.data;
        const char* const e = fqmap.data + fqmap.size;
        // the next line [b, l) without its newline; false at the end of the file (std::getline's rules: a last line without a
        // newline counts, an empty remainder does not)
        auto line = [&](const char*& b0, const char*& l0) {
            if (p >= e) return false;
            const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(e - p)));
            b0 = p;
            l0 = nl ? nl : e;
            p = nl ? nl + 1 : e;
            return true;
        };
        const char *hb, *he, *sb, *se, *pb, *pe, *qb, *qe;
        while (line(hb, he)) {
            if (he == hb) continue;
            const string h(hb, size_t(he - hb));
            if (!line(sb, se) || !line(pb, pe) || !line(qb, qe)) die("Truncated fastq record: " + h);
            if (hb[0] != '@' || se - sb != qe - qb) die("Malformed fastq record: " + h);
            Seq r;
            const size_t sp = h.find_first_of(" \t");
            r.name = h.substr(1, sp == string::npos ? string::npos : sp - 1);
            r.seq = Bytes(sb, size_t(se - sb), fqmap.keep);
            r.qual = Bytes(qb, size_t(qe - qb), fqmap.keep);
            reads.push_back(std::move(r));
        }
    }
    const int n = int(reads.size());
    if (VERBOSE) cerr << "Parsed " << n << " sequences." << endl;
    lap("fastq parsed");

    ioc_ctx* c = ctx_future.get();
    lap("context (started beside the parsing)");
    // ---- FillQualScores on the device, SortByQualScores on the host ----
    std::vector<int64_t> offs(static_cast<size_t>(n) + 1, 0);
    for (int i = 0; i < n; ++i) offs[size_t(i) + 1] = offs[size_t(i)] + int64_t(reads[size_t(i)].qual.size());
    {
        std::shared_ptr<void> qual_mem = huge_alloc(static_cast<size_t>(offs[size_t(n)]) + 1);
        uint8_t* const qual = static_cast<uint8_t*>(qual_mem.get());
        {
            const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
            std::vector<std::thread> th;
            for (unsigned t0 = 0; t0 < nt; ++t0)
                th.emplace_back([&, t0] {
                    for (int i = int(t0); i < n; i += int(nt)) memcpy(qual + offs[size_t(i)], reads[size_t(i)].qual.data(), reads[size_t(i)].qual.size());
                });
            for (auto& x : th) x.join();
        }
        std::vector<double> score(static_cast<size_t>(n)), err(static_cast<size_t>(n));
        check(c, ioc_qual_scores(c, n, offs.data(), qual, a.KmerSize, score.data(), err.data()), "quality scores");
        for (int i = 0; i < n; ++i) {
            reads[size_t(i)].score = score[size_t(i)];
            reads[size_t(i)].errorRate = err[size_t(i)];
        }
    }
    lap("quality scores (GPU)");
    std::stable_sort(reads.begin(), reads.end(), [](const Seq& x, const Seq& y) { return x.score > y.score; });
    lap("stable sort");

    const string batch_dir = a.BatchOutFolder + "/batches";
    create_outdir(a.BatchOutFolder);
    create_outdir(batch_dir);
    {
        std::ofstream tsv, sc;
        const string sorted = a.BatchOutFolder + "/sorted_reads.fastq";
        GatherFile fq;  // (the bases and qualities go from the input's mapping to the page cache in one gather per 1024 pieces)
        if (!fq.open(sorted)) die("Failed to open " + sorted + "!");
        create_file(a.BatchOutFolder + "/sorted_reads_idx.tsv", tsv);
        tsv << "Id\tPos" << endl;
        unsigned long long seek = 0;
        for (auto& r : reads) {
            if (r.score < 0) continue;
            tsv << r.name << "\t" << seek << "\n";  // (the same bytes as endl, without a write() per read)
            fq.put("@", 1);
            fq.put(r.name.data(), r.name.size());
            fq.put("\n", 1);
            fq.put(r.seq.data(), r.seq.size());
            fq.put("\n+\n", 3);
            fq.put(r.qual.data(), r.qual.size());
            fq.put("\n", 1);
            seek += r.name.size() + r.seq.size() + r.qual.size() + 6;
        }
        if (!fq.close()) die("Failed to write " + sorted + "!");
        save_sorted_idx(sorted, a.BatchOutFolder + "/sorted_reads_idx.cer");
        create_file(a.BatchOutFolder + "/scores.tsv", sc);
        for (auto& r : reads) sc << r.name << "\t" << r.score << "\n";
    }

    lap("sorted fastq, index, scores written");
    // ---- batches (main.cpp:149-199) ----
    // The batches are independent once the reads are sorted: the extraction of a batch runs on the one context (a mutex around
    // the three calls), the assembly of its records and its file are built by worker threads side by side (IOC_SORT_THREADS,
    // default 6: the 64 batches of a 2 M-read file took 29 s one after the other).
    std::mutex ctx_mu, log_mu;
    std::atomic<long long> us_dev{0}, us_asm{0}, us_save{0};  // (IOC_TRACE: summed over the batches' threads)
    auto us_now = [] { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto write_batch = [&](int start, int end, unsigned long bases, int nr) {
        // (every large buffer of a batch comes in huge pages — 0.9 GB of first touches per 31 250-read batch otherwise, six threads at
        // once — and the records are views of them: nothing is copied per read, cer.hpp)
        const int m = end - start + 1;
        std::vector<int64_t> bo(static_cast<size_t>(m) + 1, 0);
        for (int i = 0; i < m; ++i) bo[size_t(i) + 1] = bo[size_t(i)] + int64_t(reads[size_t(start + i)].seq.size());
        const size_t nb = static_cast<size_t>(bo[size_t(m)]);
        std::shared_ptr<void> seq_mem = huge_alloc(nb + 1), qual_mem = huge_alloc(nb + 1);
        uint8_t* const seq = static_cast<uint8_t*>(seq_mem.get());
        uint8_t* const qual = static_cast<uint8_t*>(qual_mem.get());
        for (int i = 0; i < m; ++i) {
            memcpy(seq + bo[size_t(i)], reads[size_t(start + i)].seq.data(), reads[size_t(start + i)].seq.size());
            memcpy(qual + bo[size_t(i)], reads[size_t(start + i)].qual.data(), reads[size_t(start + i)].qual.size());
        }
        std::vector<uint32_t> hlen(static_cast<size_t>(m));
        std::vector<double> herr(static_cast<size_t>(m));
        std::vector<int64_t> of(static_cast<size_t>(m) + 1), orv(static_cast<size_t>(m) + 1);
        std::vector<int32_t> status(static_cast<size_t>(m));
        std::shared_ptr<void> hs_mem = huge_alloc(nb + 1), hq_mem = huge_alloc(nb + 1);
        char* const hs = static_cast<char*>(hs_mem.get());
        char* const hq = static_cast<char*>(hq_mem.get());
        std::unique_lock<std::mutex> dev(ctx_mu);  // ---- the one context: extraction and its downloads, one batch at a time ----
        const long long t_dev = us_now();
        check(c, ioc_extract_minimizers(c, m, bo.data(), seq, qual, a.KmerSize, a.WindowSize, hlen.data(),
                                        herr.data(), of.data(), orv.data(), status.data()), "minimizer extraction");
        const int64_t tot = orv[size_t(m)];
        std::shared_ptr<void> mv_mem = huge_alloc((static_cast<size_t>(tot) + 1) * 4), mp_mem = huge_alloc((static_cast<size_t>(tot) + 1) * 4);
        uint32_t* const mv = static_cast<uint32_t*>(mv_mem.get());
        uint32_t* const mp = static_cast<uint32_t*>(mp_mem.get());
        check(c, ioc_extracted_download(c, mv, mp, tot + 1), "minimizer download");
        check(c, ioc_extracted_hpc_download(c, hs, hq, int64_t(nb + 1)), "hpc download");
        dev.unlock();
        const long long t_asm = us_now();
        us_dev += t_asm - t_dev;
        seq_mem.reset();
        qual_mem.reset();
        // the minimizers as the records hold them (Min, Pos, Index): one array for the batch
        std::shared_ptr<void> aos_mem = huge_alloc((static_cast<size_t>(tot) + 1) * sizeof(Minimizer));
        Minimizer* const aos = static_cast<Minimizer*>(aos_mem.get());
        Batch b;
        b.Cls.resize(size_t(m));
        for (int i = 0; i < m; ++i) {
            Seq& r = reads[size_t(start + i)];
            auto cl = std::make_shared<Cluster>();
            auto ps = std::make_shared<ProcSeq>();
            ps->Id = r.name;
            const bool lowq = (-10 * log10(r.errorRate)) <= a.MinQual;            // qualscore.cpp:56
            const bool lenok = r.seq.size() > size_t(2 * a.KmerSize) || r.seq.size() >= size_t(a.WindowSize);
            if (status[size_t(i)] == 2) die("Invalid base encountered in read " + r.name);   // RevComp throws
            if (lowq) {
                // placeholder {nullptr, nullptr, {}, {}, 0, name}
            } else if (!lenok || status[size_t(i)] == 1) {
                r.score = -1.0;  // qualscore.cpp:67, :91 (the reference's raw-only branch is unreachable without a crash)
            } else {
                ps->RawSeq.reset(new Seq(r));
                ps->HpcSeq.reset(new Seq);
                ps->HpcSeq->name = r.name;
                ps->HpcSeq->seq = Bytes(hs + bo[size_t(i)], hlen[size_t(i)], hs_mem);
                ps->HpcSeq->qual = Bytes(hq + bo[size_t(i)], hlen[size_t(i)], hq_mem);
                ps->HpcSeq->score = r.score;
                ps->HpcSeq->errorRate = herr[size_t(i)];
                auto fill = [&](Span<Minimizer>& out, int64_t b0, int64_t e0) {
                    for (int64_t t = b0; t < e0; ++t) aos[size_t(t)] = Minimizer{mv[size_t(t)], mp[size_t(t)], uint32_t(t - b0)};
                    out = Span<Minimizer>(aos + b0, size_t(e0 - b0), aos_mem);
                };
                fill(ps->Mins, of[size_t(i)], of[size_t(i) + 1]);
                fill(ps->RevMins, orv[size_t(i)], orv[size_t(i) + 1]);
                ps->MatchStrand = 1;
            }
            cl->push_back(ps);
            b.Cls[size_t(i)] = cl;
        }
        b.NrCls = m;
        b.BatchStart = uint64_t(start);
        b.BatchEnd = uint64_t(end);
        b.Depth = -1;
        b.BatchNr = nr;
        b.BatchBases = bases;
        b.SortArgs = a;
        string err;
        const long long t_save = us_now();
        us_asm += t_save - t_asm;
        if (!save_batch(b, batch_dir + "/isONbatch_" + std::to_string(nr) + ".cer", err)) die(err);
        us_save += us_now() - t_save;
        if (VERBOSE) {
            std::lock_guard<std::mutex> lk(log_mu);
            cerr << "\tWritten batch " << nr << " with " << m << " sequences and " << int(double(bases) / 1000.0) << " kilobases." << endl;
        }
    };
    struct Range {
        int start, end;
        unsigned long bases;
        int nr;
    };
    std::vector<Range> ranges;
    unsigned long batch_bases = 0;
    int batch_seqs = 0, nr_batches = 0, batch_start = 0;
    for (int i = 0; i < n; ++i) {
        batch_bases += reads[size_t(i)].seq.size();
        batch_seqs++;
        if (a.BatchSize > 0 && (batch_bases > (unsigned long)(a.BatchSize) * 1000ul || (a.BatchMaxSeq > 0 && batch_seqs >= a.BatchMaxSeq))) {
            ranges.push_back(Range{batch_start, i, batch_bases, nr_batches});
            batch_bases = 0;
            batch_seqs = 0;
            batch_start = i + 1;
            nr_batches++;
        }
    }
    if (batch_start < n) ranges.push_back(Range{batch_start, n - 1, batch_bases, nr_batches});
    {
        int nt = 6;
        if (const char* e = getenv("IOC_SORT_THREADS")) nt = std::max(1, atoi(e));
        nt = std::min<int>(nt, int(ranges.size()));
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t x = next.fetch_add(1); x < ranges.size(); x = next.fetch_add(1)) write_batch(ranges[x].start, ranges[x].end, ranges[x].bases, ranges[x].nr);
        };
        if (nt <= 1) {
            work();
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(work);
            for (auto& t : th) t.join();
        }
    }
    lap("batches extracted and written");
    if (trace)
        fprintf(stderr, "[ioc] sort:   of it, summed over the batches: device section (one at a time) %.1f ms, record assembly %.1f ms, file %.1f ms\n", us_dev.load() / 1e3,
                us_asm.load() / 1e3, us_save.load() / 1e3);
    ioc_ctx_destroy(c);
    return 0;
}

// ===================================================================================================
// cluster  (src/main.cpp:238-382 + the bookkeeping of src/cluster.cpp:67-322)
// ===================================================================================================
static int main_cluster(int argc, char** argv)
{
    static const struct option lo[] = {
        {"quiet", no_argument, 0, 'Q'}, {"version", no_argument, 0, 'V'}, {"verbose", no_argument, 0, 'v'},
        {"min-purge", no_argument, 0, 'z'}, {"min-cls-size", required_argument, 0, 'F'}, {"keep-seq", no_argument, 0, 'j'},
        {"debug", no_argument, 0, 'd'}, {"spoa-algo", required_argument, 0, 'A'}, {"mode", required_argument, 0, 'x'},
        {"help", no_argument, 0, 'h'}, {"outfile", required_argument, 0, 'o'}, {"left-batch", required_argument, 0, 'l'},
        {"right-batch", required_argument, 0, 'r'}, {0, 0, 0, 0}};
    string left_path, right_path, out_path;
    int mode = None, min_cls = -1;
    bool min_purge = false, keep_seq = false;
    int o;
    while ((o = getopt_long(argc, argv, "Vdhvo:l:r:Qx:A:zjF:", lo, nullptr)) != -1) {
        switch (o) {
            case 'o': out_path = optarg; break;
            case 'l': left_path = optarg; break;
            case 'r': right_path = optarg; break;
            case 'v': VERBOSE = true; break;
            case 'z': min_purge = true; break;
            case 'j': keep_seq = true; break;
            case 'F': min_cls = atoi(optarg); break;
            case 'x': mode = parse_mode(optarg); break;
            case 'h': cerr << "isONclust2-hip cluster -l left.cer [-r right.cer] -o out.cer [-x fast|sahlin|furious] [-F n] [-z] [-j] [-v]" << endl; exit(0);
            default: break;
        }
    }
    if (left_path.empty()) die("Specifying left input batch is mandatory!");
    if (out_path.empty()) die("Specifying output batch file is mandatory!");
    auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // HIP initialisation (~80 ms) runs beside the archive load
    auto ctx_future = std::async(std::launch::async, [] {
        const auto t0 = std::chrono::steady_clock::now();
        const bool fresh = g_srv_ctx == nullptr;
        ioc_ctx* cc = g_srv_ctx ? g_srv_ctx : make_ctx();
        if (g_served) g_srv_ctx = cc;
        if (fresh && !getenv("IOC_NO_PREWARM")) (void)ioc_ctx_prewarm(cc, 1);  // (code objects load beside the flattening and the uploads)
        return std::make_pair(cc, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    });
    Batch left, right;
    string err;
    if (!load_batch(left, left_path, err)) die(err);
    double load_ms = ms_since(t_begin);
    if (VERBOSE) {
        cerr << "Loaded input batch from " << left_path << ":" << endl;
        print_batch_info(left);
    }
    const bool single = right_path.empty();
    if (!single) {
        if (!load_batch(right, right_path, err)) die(err);
        load_ms = ms_since(t_begin);
        cerr << "Loaded input batch from " << right_path << ":" << endl;
        right.Db.clear();
        print_batch_info(right);
    } else {  // CreatePseudoBatch, serialize.cpp:29-43
        right.BatchNr = -left.BatchNr;
        right.BatchStart = left.BatchStart;
        right.BatchEnd = left.BatchEnd;
        right.BatchBases = 0;
        right.SortArgs = left.SortArgs;
        right.Depth = -1;
        right.Cls = left.Cls;
        right.NrCls = int32_t(right.Cls.size());
        left.Cls.clear();
        if (left.Depth > 0) left.Depth = -left.Depth;
        left.NrCls = 0;
        left.Db.clear();
    }
    // An output file that exists is going to be replaced: giving its pages back (50 - 90 ms for a 460 MB file in the page cache,
    // which O_TRUNC would spend between the last kernel and the first byte written) happens now, beside the GPU's start.  After
    // the loads: the output may be one of the inputs, whose mapping keeps the old file alive.
    auto unlink_old = std::async(std::launch::async, [out_path] { (void)unlink(out_path.c_str()); });
    left.SortArgs.Mode = mode;
    right.SortArgs.Mode = mode;
    if (min_cls > 0) left.SortArgs.MinClsSize = min_cls;
    if (VERBOSE && mode == None) die("Invalid clustering mode: 3");
    // ---- batch compatibility checks, cluster.cpp:70-90 ----
    {
        const CmdArgs &x = left.SortArgs, &y = right.SortArgs;
        if (!(x.KmerSize == y.KmerSize && x.WindowSize == y.WindowSize && x.MinShared == y.MinShared && x.MinQual == y.MinQual &&
              x.MappedThreshold == y.MappedThreshold && x.AlignedThreshold == y.AlignedThreshold &&
              x.MinFraction == y.MinFraction && x.MinProbNoHits == y.MinProbNoHits && x.Mode == y.Mode))
            die("The left and right batches have been sorted with different parameters! \nRefusing to carry on with clustering as results would not make sense! ");
        if (right.Depth > 0 && right.BatchStart != left.BatchEnd + 1) die("Trying to merge non-consecutive batches! Giving up!");
        if (left.Depth > 0 && right.Depth > left.Depth) die("The left input batch must have higher depth!");
    }
    const CmdArgs& a = left.SortArgs;
    const bool cons_on = left.SortArgs.ConsMaxSize > 0;  // consensus mode: frozen at sort time (-c), cluster.cpp:101
    const bool need_seq = cons_on || ((mode == Sahlin || mode == Furious));

    // ---- right batch -> ioc_batch_view ----
    const int n = int(right.Cls.size());
    std::vector<int64_t> of(static_cast<size_t>(n) + 1, 0), orv(static_cast<size_t>(n) + 1, 0), roff(static_cast<size_t>(n) + 1, 0);
    std::vector<uint32_t> raw_len(static_cast<size_t>(n) + 1, 0), hpc_len(static_cast<size_t>(n) + 1, 0);
    std::vector<double> score(static_cast<size_t>(n) + 1, -1.0), raw_err(static_cast<size_t>(n) + 1, 1.0), hpc_err(static_cast<size_t>(n) + 1, 1.0);
    std::vector<uint8_t> state(static_cast<size_t>(n) + 1, 1);
    std::vector<int32_t> nmem(static_cast<size_t>(n) + 1, 0);
    int64_t tot = 0, rtot = 0;
    auto rep_of = [&](int i) -> ProcSeq* {
        auto& e = right.Cls[size_t(i)];
        if (!e || e->empty() || !e->at(0) || !e->at(0)->RawSeq) return nullptr;
        return e->at(0).get();
    };
    for (int i = 0; i < n; ++i) {
        of[size_t(i)] = tot;
        if (ProcSeq* r = rep_of(i)) tot += int64_t(r->Mins.size());
    }
    of[size_t(n)] = tot;
    for (int i = 0; i < n; ++i) {
        orv[size_t(i)] = tot;
        if (ProcSeq* r = rep_of(i)) tot += int64_t(r->RevMins.size());
    }
    orv[size_t(n)] = tot;
    // (huge pages: 191 MB of first touches on config 2's batch)
    std::shared_ptr<void> mv_mem = huge_alloc((static_cast<size_t>(tot) + 1) * 4), mp_mem = huge_alloc((static_cast<size_t>(tot) + 1) * 4);
    uint32_t* const mv = static_cast<uint32_t*>(mv_mem.get());
    uint32_t* const mp = static_cast<uint32_t*>(mp_mem.get());
    // AoS (Min, Pos, Index) -> SoA, entries spread over the host cores
    {
        const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        std::atomic<int> bad(0);
        for (unsigned t0 = 0; t0 < nt; ++t0)
            th.emplace_back([&, t0] {
                for (int i = int(t0); i < n; i += int(nt)) {
                    ProcSeq* r = rep_of(i);
                    if (!r) continue;
                    uint32_t* v = mv + of[size_t(i)];
                    uint32_t* q = mp + of[size_t(i)];
                    const Minimizer* m = r->Mins.data();  // (packed: read where the archive has them)
                    unsigned wrong = 0;
                    for (size_t t = 0, e = r->Mins.size(); t < e; ++t) {
                        wrong |= m[t].Index ^ uint32_t(t);
                        v[t] = m[t].Min;
                        q[t] = m[t].Pos;
                    }
                    v = mv + orv[size_t(i)];
                    q = mp + orv[size_t(i)];
                    m = r->RevMins.data();
                    for (size_t t = 0, e = r->RevMins.size(); t < e; ++t) {
                        wrong |= m[t].Index ^ uint32_t(t);
                        v[t] = m[t].Min;
                        q[t] = m[t].Pos;
                    }
                    if (wrong) bad = 1;
                }
            });
        for (auto& x : th) x.join();
        if (bad) die("Minimizer Index is not the ordinal");
    }
    for (int i = 0; i < n; ++i) {
        roff[size_t(i)] = rtot;
        ProcSeq* r = rep_of(i);
        if (!r) continue;
        if (!r->HpcSeq) die("Entry without HpcSeq in the right batch");
        state[size_t(i)] = 0;
        nmem[size_t(i)] = int32_t(right.Cls[size_t(i)]->size()) - 1;
        raw_len[size_t(i)] = uint32_t(r->RawSeq->seq.size());
        hpc_len[size_t(i)] = uint32_t(r->HpcSeq->seq.size());
        score[size_t(i)] = r->RawSeq->score;
        raw_err[size_t(i)] = r->RawSeq->errorRate;
        hpc_err[size_t(i)] = r->HpcSeq->errorRate;
        if (need_seq) rtot += int64_t(r->RawSeq->seq.size());
    }
    // the raw sequences side by side (alignment modes, consensus): one huge-page buffer, filled by a few threads
    std::shared_ptr<void> rseq_mem = huge_alloc(size_t(rtot) + 1);
    char* const rseq = static_cast<char*>(rseq_mem.get());
    if (need_seq && rtot > 0) {
        const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned t0 = 0; t0 < nt; ++t0)
            th.emplace_back([&, t0] {
                for (int i = int(t0); i < n; i += int(nt))
                    if (ProcSeq* r = rep_of(i)) memcpy(rseq + roff[size_t(i)], r->RawSeq->seq.data(), r->RawSeq->seq.size());
            });
        for (auto& x : th) x.join();
    }
    roff[size_t(n)] = rtot;
    ioc_batch_view rv{};
    rv.n = n;
    rv.off_fwd = of.data();
    rv.off_rev = orv.data();
    rv.min_val = mv;
    rv.min_pos = mp;
    rv.total = tot;
    rv.raw_len = raw_len.data();
    rv.hpc_len = hpc_len.data();
    rv.score = score.data();
    rv.raw_err = raw_err.data();
    rv.hpc_err = hpc_err.data();
    rv.state = state.data();
    rv.min_qual = a.MinQual;
    rv.raw_seq = need_seq ? rseq : nullptr;
    rv.raw_off = need_seq ? roff.data() : nullptr;
    rv.n_members = nmem.data();
    rv.depth = right.Depth;
    rv.min_cls_size = a.MinClsSize;

    // ---- left batch -> ioc_left_view ----
    const int L = int(left.Cls.size());
    std::vector<double> l_hpc_err(static_cast<size_t>(L) + 1, 0.0), l_raw_err(static_cast<size_t>(L) + 1, 0.0);
    std::vector<int64_t> l_off(static_cast<size_t>(L) + 1, 0), k_offs;
    std::vector<uint32_t> keys, post;
    string lseq;
    for (int i = 0; i < L; ++i) {
        auto& cl = left.Cls[size_t(i)];
        if (!cl || cl->empty() || !cl->at(0) || !cl->at(0)->HpcSeq || !cl->at(0)->RawSeq) die("Left cluster without representative");
        l_hpc_err[size_t(i)] = cl->at(0)->HpcSeq->errorRate;
        l_raw_err[size_t(i)] = cl->at(0)->RawSeq->errorRate;
        l_off[size_t(i)] = int64_t(lseq.size());
        if (need_seq) lseq.append(cl->at(0)->RawSeq->seq.data(), cl->at(0)->RawSeq->seq.size());
    }
    l_off[size_t(L)] = int64_t(lseq.size());
    k_offs.push_back(0);
    for (auto& kv : left.Db) {
        if (kv.second.empty()) continue;
        keys.push_back(kv.first);
        kv.second.append_to(post);
        k_offs.push_back(int64_t(post.size()));
    }
    ioc_left_view lv{};
    lv.n_clusters = L;
    lv.cls_hpc_err = l_hpc_err.data();
    lv.n_keys = int64_t(keys.size());
    lv.keys = keys.data();
    lv.offs = k_offs.data();
    lv.postings = post.data();
    lv.rep_seq = need_seq ? lseq.data() : nullptr;
    lv.rep_off = need_seq ? l_off.data() : nullptr;
    lv.cls_raw_err = l_raw_err.data();

    ioc_params p{a.KmerSize, a.WindowSize, a.MinShared, mode, a.MinFraction, a.MappedThreshold, a.MinProbNoHits, a.AlignedThreshold};
    const double flatten_ms = ms_since(t_begin) - load_ms;
    auto ctx_done = ctx_future.get();
    ioc_ctx* c = ctx_done.first;
    const double ctx_ms = ctx_done.second;
    std::vector<int32_t> out_cls(static_cast<size_t>(n) + 1);
    std::vector<int8_t> out_strand(static_cast<size_t>(n) + 1);
    ioc_cluster_stats st{};
    auto t_core = std::chrono::steady_clock::now();
    // consensus mode: the graphs live in this build's POA engine (spoa is not in the reference tree); representatives
    // replaced by a consensus are collected and written into the clusters after the bookkeeping below
    struct RepEvent {
        int32_t entry = -1;
        string raw, hpc;
        char qual = '!';
        double raw_err = 0, raw_score = 0, hpc_err = 0;
        std::vector<cer::Minimizer> mins, rev;
    };
    std::map<int32_t, RepEvent> rep_events;
    ioc_poa* poa = nullptr;
    struct PoaGuard {  // (a served job that leaves through die() must not leave its engine behind)
        ioc_poa*& p;
        ~PoaGuard()
        {
            if (p && g_served) ioc_poa_destroy(p);
        }
    } poa_guard{poa};
    if (cons_on) {
        if (mode == None) die("Invalid clustering mode: 3");
        check(c, ioc_poa_create(c, 4, -8, -8, -4, -20, -1, &poa), "consensus engine");  // src/main.cpp:285-290
        auto load_side = [&](int side, const decltype(left.ConsGs)& gs, size_t limit, const char* what) {
            std::vector<int32_t> ids;
            std::vector<const uint8_t*> ptr;
            std::vector<int64_t> len;
            for (size_t i = 0; i < gs.size() && i < limit; ++i)
                if (!gs[i].empty()) {
                    ids.push_back(int32_t(i));
                    ptr.push_back(gs[i].data());
                    len.push_back(int64_t(gs[i].size()));
                }
            check(c, ioc_poa_graph_load_many(poa, side, int32_t(ids.size()), ids.data(), ptr.data(), len.data()), what);
        };
        load_side(0, left.ConsGs, size_t(L), "left consensus graph");
        load_side(1, right.ConsGs, size_t(n), "right consensus graph");
        // left clusters without a stored graph (batches clustered without consensus): seeded with the representative
        for (int i = 0; i < L; ++i)
            if (size_t(i) >= left.ConsGs.size() || left.ConsGs[size_t(i)].empty()) {
                ioc_consensus_ops seed{};
                ioc_poa_bind(poa, &seed);
                const Bytes& rs0 = left.Cls[size_t(i)]->at(0)->RawSeq->seq;
                seed.create(seed.user, 0, i, rs0.data(), int(rs0.size()));
            }
        ioc_consensus_ops ops{};
        ioc_poa_bind(poa, &ops);
        struct Ctx {
            std::map<int32_t, RepEvent>* ev;
            void* poa;
        } cbctx{&rep_events, poa};
        // the five graph operations go to the engine; rep_changed is ours (different `user`): wrap
        static Ctx* g_cb = nullptr;
        g_cb = &cbctx;
        ops.user = poa;
        ops.rep_changed = [](void*, int32_t cls, const ioc_rep_record* rec) {
            RepEvent& e = (*g_cb->ev)[cls];
            e.entry = rec->entry;
            e.raw.assign(rec->raw_seq, size_t(rec->raw_len));
            e.hpc.assign(rec->hpc_seq, size_t(rec->hpc_len));
            e.qual = rec->raw_qual;
            e.raw_err = rec->raw_err;
            e.raw_score = rec->raw_score;
            e.hpc_err = rec->hpc_err;
            e.mins.resize(size_t(rec->n_fwd));
            for (int32_t t = 0; t < rec->n_fwd; ++t) e.mins[size_t(t)] = cer::Minimizer{rec->fwd_min[t], rec->fwd_pos[t], uint32_t(t)};
            e.rev.resize(size_t(rec->n_rev));
            for (int32_t t = 0; t < rec->n_rev; ++t) e.rev[size_t(t)] = cer::Minimizer{rec->rev_min[t], rec->rev_pos[t], uint32_t(t)};
        };
        std::vector<int32_t> lsizes(static_cast<size_t>(L) + 1, 2);
        for (int i = 0; i < L; ++i) lsizes[size_t(i)] = int32_t(left.Cls[size_t(i)]->size());
        const CmdArgs& la = left.SortArgs;
        ioc_consensus_args ca{la.ConsMinSize, la.ConsMaxSize, la.ConsPeriod, left.Depth, lsizes.data()};
        check(c, ioc_cluster_consensus(c, &p, table_path().c_str(), L > 0 ? &lv : nullptr, &rv, &ca, &ops, out_cls.data(),
                                       out_strand.data(), &st),
              "clustering (consensus mode)");
    } else {
        check(c, ioc_cluster_merge(c, &p, table_path().c_str(), L > 0 ? &lv : nullptr, &rv, out_cls.data(), out_strand.data(), &st),
              "clustering");
    }
    double core_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_core).count();
    // a one-shot process gives the aligner's arena back NOW, beside its own bookkeeping and writing: the driver wipes released
    // VRAM before anybody gets it again, and the next `cluster` process of a pipeline would wait for that in its own first
    // large allocation (0.27 ms or 400 ms for the same 8 GB hipMalloc: profiles/r05_cli_breakdown.txt)
    if (!g_served && (mode == Sahlin || mode == Furious)) (void)ioc_ctx_trim(c);
    const double trim_ms = ms_since(t_core) - core_ms;

    // ---- bookkeeping of the loop, cluster.cpp:115-310 ----
    // representative copies of the fresh reads that open clusters (cluster.cpp:181-199): the large fields are immutable views
    // (cer.hpp), so a copy shares them — only the names are new
    std::vector<std::shared_ptr<ProcSeq>> rep_copy(static_cast<size_t>(n));
    {
        std::vector<int> fresh;
        int next_id = int(left.Cls.size());
        std::vector<int> new_id(static_cast<size_t>(n), -1);
        for (int i = 0; i < n; ++i)
            if (out_cls[size_t(i)] >= 0 && out_cls[size_t(i)] == next_id) {
                new_id[size_t(i)] = next_id++;
                if (right.Cls[size_t(i)] && right.Cls[size_t(i)]->size() == 1 && rep_of(i)) fresh.push_back(i);
            }
        for (const int i : fresh) {
            ProcSeq* r = rep_of(i);
            auto rep = std::make_shared<ProcSeq>();
            rep->RawSeq.reset(new Seq(*r->RawSeq));
            rep->HpcSeq.reset(new Seq(*r->HpcSeq));
            rep->Mins = r->Mins;
            rep->RevMins = r->RevMins;
            rep->MatchStrand = r->MatchStrand;
            rep->Id = r->Id;
            const string nm = "rep_" + std::to_string(left.BatchNr) + "_" + std::to_string(new_id[size_t(i)]);
            rep->RawSeq->name = nm;
            rep->HpcSeq->name = nm;
            rep_copy[size_t(i)] = rep;
        }
    }
    for (int i = 0; i < n; ++i) {
        auto& entry = right.Cls[size_t(i)];
        ProcSeq* r = rep_of(i);
        if (out_cls[size_t(i)] < 0) {
            if (r && r->RawSeq->score >= 0) {
                // gates that mark the read as unusable (cluster.cpp:148-160)
                if (r->RawSeq->seq.size() < size_t(2 * a.KmerSize) || r->HpcSeq->seq.size() < size_t(2 * a.KmerSize) ||
                    (-10 * log10(r->RawSeq->errorRate)) <= a.MinQual)
                    r->RawSeq->score = -1.0;
            }
            continue;
        }
        const int best = out_cls[size_t(i)];
        if (best == int(left.Cls.size())) {  // opens a new cluster (cluster.cpp:177-222)
            if (entry->size() == 1) {
                if (!rep_copy[size_t(i)]) die("Inconsistent cluster id from the device path");
                entry->insert(entry->begin(), rep_copy[size_t(i)]);
            }
            left.Cls.push_back(entry);
            left.NrCls++;
        } else {  // joins (cluster.cpp:223-261)
            if (best > int(left.Cls.size())) die("Inconsistent cluster id from the device path");
            for (auto& s : *entry) {
                if (!s) die("Null pointer in read array");
                if (out_strand[size_t(i)] == -1) {
                    if (s->MatchStrand == 1) s->MatchStrand = -1;
                    else if (s->MatchStrand == -1) s->MatchStrand = 1;
                    else die("Invalid match strand!");
                }
                s->Mins.clear();
                s->RevMins.clear();
                if (!keep_seq) {
                    s->RawSeq.reset();
                    s->HpcSeq.reset();
                }
            }
            auto& dst = *left.Cls[size_t(best)];
            size_t from = entry->size() > 1 ? 1 : 0;
            for (size_t t = from; t < entry->size(); ++t) dst.push_back((*entry)[t]);
        }
    }
    left.Depth++;
    left.BatchEnd = right.BatchEnd;
    left.BatchBases += right.BatchBases;
    // MinDB after AddMinimizers of every new representative
    {
        int64_t nk = 0, np = 0;
        check(c, ioc_index_export(c, &nk, &np, nullptr, nullptr, nullptr), "index export");
        std::vector<uint32_t> ek(static_cast<size_t>(nk) + 1);
        auto ep = std::make_shared<std::vector<uint32_t>>(static_cast<size_t>(np) + 1);  // one flat array; the lists are views of it
        std::vector<int64_t> eo(static_cast<size_t>(nk) + 2);
        check(c, ioc_index_export(c, &nk, &np, ek.data(), eo.data(), ep->data()), "index export");
        left.Db.clear();
        left.Db.reserve(size_t(nk));
        for (int64_t i = 0; i < nk; ++i)
            left.Db.emplace_back(ek[size_t(i)], Span<uint32_t>(ep->data() + eo[size_t(i)], size_t(eo[size_t(i) + 1] - eo[size_t(i)]), ep));
    }
    if (VERBOSE) {
        cerr << "Finished clustering!" << endl;
        cerr << "Alignment invocation count: " << st.n_aln_invoked << " (" << (n ? double(st.n_aln_invoked) / n * 100 : 0.0) << "%)" << endl;
        cerr << "Consensus invocation count: 0 (0%)" << endl;
        unsigned cnt = 0;
        for (auto& cl : left.Cls) cnt += cl->size() > 1;
        cerr << "Number of clusters larger than 1: " << cnt << endl;
        cerr << "Output batch statistics:" << endl;
        print_batch_info(left);
    }
    left.LeftLeaf = left_path;
    left.RightLeaf = right_path;
    if (min_purge) {
        cerr << "Purging minimizer database in output batch!" << endl;
        left.Db.clear();
    }
    left.NrConsGs = left.Cls.size();
    left.ConsGs.clear();
    if (cons_on) {
        // UpdateClusterConsensus' effect on the representatives (consensus.cpp:93-124), last event per cluster
        for (auto& kv : rep_events) {
            if (kv.first < 0 || size_t(kv.first) >= left.Cls.size()) die("Inconsistent cluster id from the consensus path");
            auto& rep = left.Cls[size_t(kv.first)]->at(0);
            const RepEvent& e = kv.second;
            const string nm = "cons_" + std::to_string(left.BatchNr) + "_" + std::to_string(e.entry);
            rep->RawSeq.reset(new Seq);
            rep->RawSeq->name = nm;
            rep->RawSeq->seq = e.raw;
            rep->RawSeq->qual = string(e.raw.size(), e.qual);
            rep->RawSeq->score = e.raw_score;
            rep->RawSeq->errorRate = e.raw_err;
            rep->HpcSeq.reset(new Seq);
            rep->HpcSeq->name = nm;
            rep->HpcSeq->seq = e.hpc;
            rep->HpcSeq->qual = string(e.hpc.size(), e.qual);
            rep->HpcSeq->score = e.hpc_err * double(e.hpc.size());
            rep->HpcSeq->errorRate = e.hpc_err;
            rep->Mins = e.mins;
            rep->RevMins = e.rev;
        }
        left.ConsGs.resize(left.Cls.size());
        for (size_t i = 0; i < left.Cls.size(); ++i) {
            const int64_t sz = ioc_poa_graph_save(poa, 0, int(i), nullptr, 0);
            if (sz <= 0) continue;
            left.ConsGs[i].resize(size_t(sz));
            if (ioc_poa_graph_save(poa, 0, int(i), left.ConsGs[i].data(), sz) != sz) die("Failed to serialize a consensus graph");
        }
        if (VERBOSE) cerr << "Consensus invocation count: " << st.n_cons_invoked << endl;
    }
    const double book_ms = ms_since(t_core) - core_ms - trim_ms;
    unlink_old.wait();
    auto t_save = std::chrono::steady_clock::now();
    if (!save_batch(left, out_path, err)) die(err);
    const double save_ms = ms_since(t_save);
    double cli_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (VERBOSE) cerr << "Output batch written to: " << out_path << endl;
    if (getenv("ISONCLUST2_STATS_JSON"))
        fprintf(stderr,
                "{\"entries\": %d, \"clusters\": %lld, \"core_ms\": %.3f, \"cli_ms\": %.3f, \"resolve_sweeps\": %d, "
                "\"load_ms\": %.3f, \"flatten_ms\": %.3f, \"ctx_ms\": %.3f, \"trim_ms\": %.3f, \"bookkeeping_ms\": %.3f, \"save_ms\": %.3f, "
                "\"t_begin_mono_ms\": %.3f, \"t_end_mono_ms\": %.3f}\n",
                n, (long long)st.n_clusters, core_ms, cli_ms, st.resolve_iters, load_ms, flatten_ms, ctx_ms, trim_ms, book_ms, save_ms,
                // (CLOCK_MONOTONIC, the clock of Python's time.monotonic(): a harness can tell the time before main and after _exit)
                std::chrono::duration<double, std::milli>(t_begin.time_since_epoch()).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
    if (poa && !g_served && getenv("IOC_TRACE")) ioc_poa_destroy(poa);  // (prints the engine's counters)
    fflush(nullptr);
    if (g_served) return 0;  // (the resident process keeps its context; the batch records go with this frame)
    // the output is on disk: leave without unwinding a gigabyte of host structures and the HIP runtime
    std::cout.flush();
    cerr.flush();
    if (getenv("IOC_CLI_CLEAN_EXIT")) exit(0);  // (profilers write their results from exit handlers)
    _exit(0);
}

// ===================================================================================================
// dump  (src/main.cpp:204-236, 430-453; src/output.cpp:151-275)   and   info (src/main.cpp:384-399)
// ===================================================================================================
static int main_dump(int argc, char** argv)
{
    static const struct option lo[] = {{"verbose", no_argument, 0, 'v'}, {"debug", no_argument, 0, 'd'}, {"help", no_argument, 0, 'h'},
                                       {"outdir", required_argument, 0, 'o'}, {"index", required_argument, 0, 'i'}, {0, 0, 0, 0}};
    string outdir, index;
    int o;
    while ((o = getopt_long(argc, argv, "dhvo:i:", lo, nullptr)) != -1) {
        switch (o) {
            case 'o': outdir = optarg; break;
            case 'i': index = optarg; break;
            case 'v': VERBOSE = true; break;
            case 'h': cerr << "isONclust2-hip dump -i sorted_reads_idx.cer -o outdir final.cer" << endl; exit(0);
            default: break;
        }
    }
    if (optind >= argc) die("No input batch specified!");
    if (outdir.empty()) die("Specifying output directory is mandatory!");
    if (index.empty()) die("Specifying the sorted read index is mandatory!");
    Batch b;
    string err, fastq;
    const bool trace = getenv("IOC_TRACE") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ioc] dump: %-44s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_lap).count());
        t_lap = t1;
    };
    if (!load_batch(b, argv[optind], err)) die(err);
    lap("batch loaded");
    if (!load_sorted_idx(fastq, index)) die("Failed to load index " + index);
    create_outdir(outdir);
    b.Db.clear();
    // SortClustersBySize, cluster.cpp:570-580 (std::sort, same comparator)
    std::sort(b.Cls.begin(), b.Cls.end(), [](const std::shared_ptr<Cluster> x, const std::shared_ptr<Cluster> y) {
        if (x->size() == y->size()) return x->at(0)->RawSeq->score > y->at(0)->RawSeq->score;
        return x->size() > y->size();
    });
    {
        std::ofstream bi;
        create_file(outdir + "/batch_info.tsv", bi);
        int ncls = 0, nnt = 0;
        for (auto& c : b.Cls)
            if (c->at(0)->RawSeq && c->at(0)->RawSeq->score > -1) {
                ncls++;
                nnt += c->size() > 2;
            }
        bi << "Name\tValue\nBatchNumber\t" << b.BatchNr << "\nBatchStart\t" << b.BatchStart << "\nBatchEnd\t" << b.BatchEnd << "\nDepth\t"
           << b.Depth << "\nNrBases\t" << b.BatchBases << "\nNrClusters\t" << ncls << "\nNrNontrivialCls\t" << nnt << "\nMinDBsize\t0\n";
    }
    struct IdInfo {
        unsigned cls;
        int strand;
    };
    std::unordered_map<string, IdInfo> id2cls;
    {
        std::ofstream info;
        create_file(outdir + "/clusters_info.tsv", info);
        create_outdir(outdir + "/cluster_fastq");
        info << "ClusterId\tSize" << endl;
        unsigned i = 0;
        for (auto& cl : b.Cls) {
            info << i << "\t" << cl->size() - 1 << endl;
            for (auto& m : *cl) id2cls[m->Id] = IdInfo{i, m->MatchStrand};
            i++;
        }
    }
    auto revcomp = [](string s) {
        std::reverse(s.begin(), s.end());
        for (auto& ch : s) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
        return s;
    };
    {
        std::ofstream cons;
        create_file(outdir + "/cluster_cons.fq", cons);
        for (size_t i = 0; i < b.Cls.size(); ++i) {
            auto& rep = b.Cls[i]->at(0);
            if (!rep->RawSeq) die("Null pointer instead of cluster rep sequence at index: " + std::to_string(i));
            if (rep->RawSeq->score < 0) continue;
            string seq = rep->RawSeq->seq.str();
            if (rep->MatchStrand == -1) seq = revcomp(seq);
            cons << "@cluster_" << i << " origin=" << rep->RawSeq->name << ":" << rep->MatchStrand << " length=" << seq.size()
                 << " size=" << b.Cls[i]->size() - 1 << "\n" << seq << "\n+\n" << rep->RawSeq->qual << "\n";
        }
    }
    // The sorted FASTQ is read out of a mapping; a read that keeps its strand goes into its cluster's file as the four lines it
    // is in the mapping (one piece of a gather), a read of the other strand as a reverse-complemented copy; the cluster files
    // are written side by side by a few threads.  (src/output.cpp:225-275 reads and writes record by record; the per-cluster
    // strings of round 4 held the whole 8 GB of configs[4] in anonymous memory.)
    lap("read ids, info files, consensus fastq");
    MappedFile fqmap;
    {
        string merr;
        if (!map_file(fastq, fqmap, merr)) die("Failed to open " + fastq + "!");
    }
    lap("sorted fastq mapped");
    std::ofstream tsv;
    create_file(outdir + "/clusters.tsv", tsv);
    tsv << "ClusterId\tStrand\tRead" << endl;
    struct Piece {  // a read's four lines in the mapping
        const char *hb, *sb, *pb, *qb, *qe;  // header, sequence, separator, qualities; he = sb - 1, se = pb - 1, pe = qb - 1
        bool flip;      // MatchStrand -1: sequence reverse-complemented, qualities reversed
        bool whole;     // the record ends with its newline: it can go out as it lies
    };
    std::unordered_map<unsigned, std::vector<Piece>> per_cluster;
    {
        const char* p = fqmap.data;
        const char* const e = fqmap.data + fqmap.size;
        auto line = [&](const char*& b0, const char*& l0) {
            if (p >= e) return false;
            const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(e - p)));
            b0 = p;
            l0 = nl ? nl : e;
            p = nl ? nl + 1 : e;
            return true;
        };
        const char *hb, *he, *sb, *se, *pb, *pe, *qb, *qe;
        while (line(hb, he) && line(sb, se) && line(pb, pe) && line(qb, qe)) {
            if (he == hb) continue;  // (std::string::substr(1) of an empty header throws in the reference's reader; nothing to keep here)
            const string id(hb + 1, size_t(he - hb - 1));
            auto it = id2cls.find(id);
            if (it == id2cls.end()) continue;
            tsv << it->second.cls << "\t" << it->second.strand << "\t" << id << "\n";
            per_cluster[it->second.cls].push_back(Piece{hb, sb, pb, qb, qe, it->second.strand == -1, p == qe + 1});
        }
    }
    lap("sorted fastq walked, clusters.tsv");
    {
        std::vector<const std::pair<const unsigned, std::vector<Piece>>*> jobs;
        for (auto& kv : per_cluster) jobs.push_back(&kv);
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned t0 = 0; t0 < nt; ++t0)
            th.emplace_back([&] {
                char comp[256];
                for (int x = 0; x < 256; ++x) comp[x] = char(x);
                comp[int('A')] = 'T', comp[int('C')] = 'G', comp[int('G')] = 'C', comp[int('T')] = 'A';
                string stage;  // the flipped records of ONE cluster (and a last record without its newline), built by this thread
                for (size_t x = next.fetch_add(1); x < jobs.size(); x = next.fetch_add(1)) {
                    GatherFile f;
                    if (!f.open(outdir + "/cluster_fastq/" + std::to_string(jobs[x]->first) + ".fq")) {
                        failed = 1;
                        continue;
                    }
                    size_t need = 0;
                    for (auto& pc : jobs[x]->second)
                        if (pc.flip || !pc.whole) need += size_t(pc.qe - pc.hb) + 2;
                    stage.clear();
                    stage.reserve(need);  // (the gather points into it: it must not move before close())
                    for (auto& pc : jobs[x]->second) {
                        if (!pc.flip && pc.whole) {
                            f.put(pc.hb, size_t(pc.qe + 1 - pc.hb));
                            continue;
                        }
                        const size_t at = stage.size();
                        stage.append(pc.hb, size_t(pc.sb - pc.hb));            // header line with its newline
                        const char* se = pc.pb - 1;
                        if (pc.flip) {
                            for (const char* c = se; c != pc.sb;) stage += comp[(unsigned char)*--c];
                            stage += '\n';
                            stage.append(pc.pb, size_t(pc.qb - pc.pb));        // separator line with its newline
                            stage.append(std::reverse_iterator<const char*>(pc.qe), std::reverse_iterator<const char*>(pc.qb));
                        } else {
                            stage.append(pc.sb, size_t(pc.qe - pc.sb));
                        }
                        stage += '\n';
                        f.put(stage.data() + at, stage.size() - at);
                    }
                    if (!f.close()) failed = 1;
                }
            });
        for (auto& x : th) x.join();
        if (failed) die("Failed to write the cluster FASTQ files!");
    }
    lap("cluster fastq files written");
    if (VERBOSE) cerr << "Dump complete." << endl;
    return 0;
}

static int main_info(int argc, char** argv)
{
    if (argc < 2 || string(argv[1]) == "-h") {
        cerr << "isONclust2-hip info batch.cer" << endl;
        exit(0);
    }
    Batch b;
    string err;
    if (!load_batch(b, argv[1], err)) die(err);
    cerr << "Loaded batch from " << argv[1] << ":" << endl;
    print_batch_info(b);
    return 0;
}

// the tiny batch whose byte image is spelled out twice: in main_selftest below and — independently, from SURVEY.md App. B and
// the serialize() member orders — in tests/test_cer_golden.py (`isONclust2-hip golden <path>` writes it)
static void make_golden_batch(Batch& g)
{
    g.BatchNr = 1;
    g.BatchStart = 2;
    g.BatchEnd = 3;
    g.BatchBases = 4;
    g.TotalReads = 5;
    g.NrCls = 1;
    g.SortArgs.InFastq = "a";
    g.SortArgs.BatchOutFolder = "b";
    g.SortArgs.Mode = Fast;
    g.LeftLeaf = "L";
    g.RightLeaf = "";
    g.Depth = -1;
    g.Db = {{9u, {0u}}};
    auto cl = std::make_shared<Cluster>();
    auto ps = std::make_shared<ProcSeq>();
    ps->RawSeq.reset(new Seq{"n", "AC", "II", 1.5, 0.25});
    ps->Mins = std::vector<Minimizer>{{1, 2, 3}};
    ps->MatchStrand = 1;
    ps->Id = "i";
    cl->push_back(ps);
    g.Cls.push_back(cl);
    g.NrConsGs = 1;
}

// .cer round trip on a synthetic batch (host only; used by the CPU test-suite)
static int main_selftest(int argc, char** argv)
{
    const string path = argc > 1 ? argv[1] : "/tmp/isonclust2_selftest.cer";
    Batch b;
    b.BatchNr = 3;
    b.BatchStart = 10;
    b.BatchEnd = 12;
    b.BatchBases = 12345;
    b.NrCls = 3;
    b.Depth = 1;
    b.SortArgs.KmerSize = 13;
    b.SortArgs.WindowSize = 20;
    b.SortArgs.Mode = Fast;
    b.SortArgs.InFastq = "reads.fq";
    b.LeftLeaf = "a.cer";
    b.RightLeaf = "";
    b.Db = {{7u, {0u, 2u}}, {0xFFFFFFFFu, {1u}}, {99u, {}}};
    for (int c = 0; c < 3; ++c) {
        auto cl = std::make_shared<Cluster>();
        for (int m = 0; m <= c; ++m) {
            auto ps = std::make_shared<ProcSeq>();
            ps->Id = "r" + std::to_string(c) + "_" + std::to_string(m);
            ps->MatchStrand = m % 2 ? -1 : 1;
            if (m == 0) {
                ps->RawSeq.reset(new Seq{ps->Id, "ACGTACGT", "IIIIIIII", 7.5, 0.01});
                ps->HpcSeq.reset(new Seq{ps->Id, "ACGT", "IIII", 7.5, 0.02});
                ps->Mins = std::vector<Minimizer>{{1, 0, 0}, {5, 3, 1}};
                ps->RevMins = std::vector<Minimizer>{{3, 6, 0}};
            }
            cl->push_back(ps);
        }
        b.Cls.push_back(cl);
    }
    b.Cls.push_back(nullptr);
    b.NrConsGs = 3;
    string err;
    if (!save_batch(b, path, err)) die(err);
    Batch r;
    if (!load_batch(r, path, err)) die(err);
    bool ok = r.BatchNr == 3 && r.BatchStart == 10 && r.BatchEnd == 12 && r.BatchBases == 12345 && r.NrCls == 3 && r.Depth == 1 &&
              r.SortArgs.KmerSize == 13 && r.SortArgs.WindowSize == 20 && r.SortArgs.Mode == Fast && r.SortArgs.InFastq == "reads.fq" &&
              r.SortArgs.MinFraction == 0.8 && r.LeftLeaf == "a.cer" && r.Db.size() == 3 && r.Db[0].first == 7u &&
              r.Db[0].second == std::vector<uint32_t>({0u, 2u}) && r.Db[2].first == 0xFFFFFFFFu && r.Cls.size() == 4 && !r.Cls[3] &&
              r.NrConsGs == 3;
    for (int c = 0; ok && c < 3; ++c) {
        ok = r.Cls[size_t(c)] && int(r.Cls[size_t(c)]->size()) == c + 1;
        for (int m = 0; ok && m <= c; ++m) {
            auto& x = (*r.Cls[size_t(c)])[size_t(m)];
            auto& y = (*b.Cls[size_t(c)])[size_t(m)];
            ok = x->Id == y->Id && x->MatchStrand == y->MatchStrand && bool(x->RawSeq) == bool(y->RawSeq) &&
                 x->Mins.size() == y->Mins.size();
            if (ok && x->RawSeq)
                ok = x->RawSeq->seq == "ACGTACGT" && x->HpcSeq->errorRate == 0.02 && x->Mins[1].Pos == 3 && x->RevMins[0].Min == 3;
        }
    }
    // a truncated file must be rejected, not crash
    {
        std::ifstream in(path, std::ios::binary);
        string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        std::ofstream out(path + ".trunc", std::ios::binary);
        out.write(all.data(), std::streamsize(all.size() / 2));
        out.close();
        Batch t;
        ok = ok && !load_batch(t, path + ".trunc", err);
        remove((path + ".trunc").c_str());
    }
    // every prefix of the file, a byte flipped at every offset, and crafted 64-bit counts (which must not wrap the
    // bounds checks) are rejected or loaded, never a crash: run under ASan / UBSan by tools/run_sanitizers.sh
    {
        std::ifstream in(path, std::ios::binary);
        const string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        auto try_load = [&](const string& bytes) {
            std::ofstream out(path + ".var", std::ios::binary);
            out.write(bytes.data(), std::streamsize(bytes.size()));
            out.close();
            Batch t;
            string e2;
            return load_batch(t, path + ".var", e2);
        };
        for (size_t cut = 0; ok && cut < all.size(); ++cut) ok = !try_load(all.substr(0, cut));
        size_t loaded = 0;
        for (size_t at = 0; at < all.size(); ++at) {
            string v = all;
            v[at] = char(v[at] ^ 0xFF);
            loaded += try_load(v);
        }
        const uint64_t huge[3] = {0x4000000000000001ull, 0xFFFFFFFFFFFFFFFFull, 0x1555555555555556ull};  // x 4 and x 12 wrap
        for (size_t at = 0; ok && at + 8 <= all.size(); ++at)
            for (uint64_t h : huge) {
                string v = all;
                memcpy(&v[at], &h, 8);
                (void)try_load(v);
            }
        remove((path + ".var").c_str());
        cerr << "selftest: " << all.size() << " prefixes rejected, " << loaded << " single-byte variants still loadable" << endl;
    }
    // byte-level image of a minimal batch with consensus off, spelled out field by field under cereal's binary
    // conventions (cereal itself is absent from the reference tree: layout unverified against a
    // reference-written file, pinned here against silent drift)
    {
        Batch g;
        make_golden_batch(g);
        string want;
        auto p32 = [&](int32_t v) { want.append(reinterpret_cast<const char*>(&v), 4); };
        auto pu32 = [&](uint32_t v) { want.append(reinterpret_cast<const char*>(&v), 4); };
        auto p64 = [&](uint64_t v) { want.append(reinterpret_cast<const char*>(&v), 8); };
        auto pf = [&](double v) { want.append(reinterpret_cast<const char*>(&v), 8); };
        auto pstr = [&](const string& x) { p64(x.size()); want += x; };
        p32(1); p64(2); p64(3); p64(4); p32(5); p32(1);                                   // BatchNr .. NrCls
        want.push_back(0); want.push_back(0); pstr("a");                                    // Verbose, Debug, InFastq
        for (int32_t v : {11, 50000, 30000, 15, 5, 50, -150, 500, 3}) p32(v);               // KmerSize .. MinClsSize
        for (double v : {7.0, 0.65, 0.2, 0.8, 0.1}) pf(v);                                  // MinQual .. MinProbNoHits
        pstr("b"); p32(Fast);                                                               // BatchOutFolder, Mode
        pstr("L"); pstr(""); p32(-1);                                                       // LeftLeaf, RightLeaf, Depth
        p64(1); pu32(9); p64(1); pu32(0);                                                   // MinDB: 1 key -> 1 posting
        p64(1); pu32(0x80000001u); p64(1); pu32(0x80000002u);                               // Cls: shared_ptr ids, first occurrence
        want.push_back(1); pstr("n"); pstr("AC"); pstr("II"); pf(1.5); pf(0.25);            // RawSeq (valid)
        want.push_back(0);                                                                  // HpcSeq (null)
        p64(1); pu32(1); pu32(2); pu32(3); p64(0); p32(1); pstr("i");                        // Mins, RevMins, MatchStrand, Id
        p64(1); want.push_back(0);                                                          // ConsGs: one null graph
        if (!save_batch(g, path, err)) die(err);
        std::ifstream in(path, std::ios::binary);
        const string got((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (got != want) {
            cerr << "selftest: byte image differs from the spelled-out layout (" << got.size() << " vs " << want.size() << " bytes)" << endl;
            ok = false;
        }
        // a non-null graph that is not this build's blob is refused with a message, not read as garbage
        string v = want;
        v.back() = 1;
        p64(8);
        v.append(want.end() - 8, want.end());
        v += "SPOAGRPH";
        {
            std::ofstream out(path, std::ios::binary);
            out.write(v.data(), std::streamsize(v.size()));
        }
        Batch t;
        string e2;
        if (load_batch(t, path, e2) || e2.find("not in this build's format") == string::npos) {
            cerr << "selftest: foreign graph blob not refused: " << e2 << endl;
            ok = false;
        }
        // ... and so is a graph of the builds before the edge weights changed (IOCPOA1)
        v.replace(v.size() - 8, 8, string("IOCPOA1\0", 8));
        {
            std::ofstream out(path, std::ios::binary);
            out.write(v.data(), std::streamsize(v.size()));
        }
        if (load_batch(t, path, e2) || e2.find("earlier build") == string::npos) {
            cerr << "selftest: graph blob of an earlier build not refused: " << e2 << endl;
            ok = false;
        }
    }
    remove(path.c_str());
    cerr << (ok ? "selftest ok" : "selftest FAILED") << endl;
    return ok ? 0 : 1;
}

// ===================================================================================================
// serve: the resident worker behind `cluster`
//
// A one-shot `cluster` process on config 2's batch spends 0.2 s starting the HIP runtime (0.06 s on a quiet card: a KFD process
// created right after another one's exit first waits for that one's teardown — the steady state of a batch-and-merge
// pipeline), 0.04 - 0.1 s in first-use costs (code objects, first allocations) and 0.1 s leaving, for 0.03 - 0.06 s of kernels
// (profiles/r05_cli_breakdown.txt).  So `cluster` hands its job to a worker process that stays: the first call of a pipeline
// starts it (fork + exec of this binary, before anything here has touched the GPU), later calls find it warm.  The command line,
// the files and the messages are the one-shot command's: the client sends its arguments, its working directory, its
// IOC_* / ISONCLUST2_* environment and its own stdout / stderr descriptors (SCM_RIGHTS), the worker runs main_cluster there on its
// resident context and answers with the exit code.  A worker takes one job at a time; concurrent callers each get their own
// (slots, chosen by a file lock the caller holds for the length of its job), a worker leaves after ISONCLUST2_SERVE_IDLE_S
// seconds (default 15) without a job.  ISONCLUST2_SERVE=0, or any failure to reach or start a worker: the job runs in this
// process, as before.  `isONclust2-hip serve stop` ends the callers' idle workers.
// ===================================================================================================
namespace serve {

static const uint32_t MAGIC = 0x31435349u;  // "ISC1"

static string dir_path()
{
    const char* base = getenv("ISONCLUST2_SERVE_DIR");
    return base ? string(base) : "/tmp/isonclust2-hip-" + std::to_string(unsigned(getuid()));
}
static int device()
{
    const char* e = getenv("ISONCLUST2_DEVICE");
    return e ? atoi(e) : 0;
}
// (a worker of another build — the binary or the library rebuilt since it started — must not answer for this one: the names
// carry a stamp of both files, the old worker idles out)
static string build_stamp()
{
    static string stamp;
    if (!stamp.empty()) return stamp;
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const char* path) {
        struct stat sb;
        if (stat(path, &sb) != 0) return;
        for (unsigned long long v : {(unsigned long long)sb.st_ino, (unsigned long long)sb.st_size, (unsigned long long)sb.st_mtim.tv_sec, (unsigned long long)sb.st_mtim.tv_nsec})
            h = (h ^ v) * 1099511628211ull;
    };
    mix("/proc/self/exe");
    Dl_info di;
    if (dladdr(reinterpret_cast<void*>(&ioc_ctx_create), &di) && di.dli_fname) mix(di.dli_fname);
    // (... nor a worker that sees other devices than the caller: the runtime reads these once, when the worker starts)
    for (const char* nm : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"}) {
        const char* v = getenv(nm);
        for (const char* q = v ? v : "\x01"; *q; ++q) h = (h ^ (unsigned long long)(unsigned char)*q) * 1099511628211ull;
        h = (h ^ 0xFFull) * 1099511628211ull;
    }
    char buf[20];
    snprintf(buf, sizeof(buf), "%08llx", h & 0xFFFFFFFFull);
    return stamp = buf;
}
static string slot_base(int slot) { return dir_path() + "/" + build_stamp() + "d" + std::to_string(device()) + "s" + std::to_string(slot); }

static bool write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) return false;
        c += w;
        n -= size_t(w);
    }
    return true;
}
static bool read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = ::recv(fd, c, n, 0);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        c += r;
        n -= size_t(r);
    }
    return true;
}
static void put_str(string& b, const string& s)
{
    const uint32_t n = uint32_t(s.size());
    b.append(reinterpret_cast<const char*>(&n), 4);
    b += s;
}

// ---- the worker ----
static int worker(int slot)
{
    const string base = slot_base(slot), sock = base + ".sock";
    // one worker per slot: the lock lives as long as this process
    const int lk = open((base + ".worker").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (lk < 0) return 0;
    {   // (a caller probing whether a worker is alive holds this lock for an instant: try a few times before concluding that
        // another worker has the slot)
        int got = -1;
        for (int t = 0; t < 40 && got != 0; ++t) {
            got = flock(lk, LOCK_EX | LOCK_NB);
            if (got != 0) usleep(500);
        }
        if (got != 0) return 0;
    }
    const int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un sa{};
    sa.sun_family = AF_UNIX;
    if (ls < 0 || sock.size() >= sizeof(sa.sun_path)) return 1;
    memcpy(sa.sun_path, sock.c_str(), sock.size() + 1);
    (void)unlink(sock.c_str());
    if (bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0 || listen(ls, 4) != 0) return 1;
    double idle_s = 15.0;
    if (const char* e = getenv("ISONCLUST2_SERVE_IDLE_S")) idle_s = std::max(0.1, atof(e));
    g_served = true;
    std::vector<string> applied;  // environment names the last job set here
    // (what the process that happened to start this worker had in ITS environment is not the next caller's: a job runs under
    // the IOC_* / ISONCLUST2_* variables its own caller sends, nothing else of that kind)
    for (char** e = environ; e && *e; ++e)
        if ((strncmp(*e, "IOC_", 4) == 0 || strncmp(*e, "ISONCLUST2_", 11) == 0) && strncmp(*e, "ISONCLUST2_SERVE_", 17) != 0) {
            const char* eq = strchr(*e, '=');
            if (eq) applied.emplace_back(*e, size_t(eq - *e));
        }
    for (;;) {
        pollfd pf{ls, POLLIN, 0};
        const int pr = poll(&pf, 1, int(idle_s * 1000.0));
        if (pr < 0 && errno == EINTR) continue;
        if (pr <= 0) break;  // idle: leave
        const int cs = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
        if (cs < 0) continue;
        // header + the caller's stdout / stderr
        uint32_t hdr[2] = {0, 0};
        int fds[2] = {-1, -1};
        {
            iovec io{hdr, sizeof(hdr)};
            alignas(cmsghdr) char cbuf[CMSG_SPACE(sizeof(fds))];
            msghdr mh{};
            mh.msg_iov = &io;
            mh.msg_iovlen = 1;
            mh.msg_control = cbuf;
            mh.msg_controllen = sizeof(cbuf);
            ssize_t r;
            do r = recvmsg(cs, &mh, MSG_CMSG_CLOEXEC); while (r < 0 && errno == EINTR);
            if (r == ssize_t(sizeof(hdr)))
                for (cmsghdr* cm = CMSG_FIRSTHDR(&mh); cm; cm = CMSG_NXTHDR(&mh, cm))
                    if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS && cm->cmsg_len == CMSG_LEN(sizeof(fds))) memcpy(fds, CMSG_DATA(cm), sizeof(fds));
        }
        bool quit = false;
        int rc = 1;
        if (hdr[0] == MAGIC && hdr[1] == 0xFFFFFFFFu) {
            quit = true;
            rc = 0;
        } else if (hdr[0] == MAGIC && hdr[1] <= (64u << 20) && fds[0] >= 0 && fds[1] >= 0) {
            string body(hdr[1], '\0');
            if (read_all(cs, &body[0], body.size())) {
                std::vector<string> args, env;
                size_t at = 0;
                auto get = [&](std::vector<string>& dst) {
                    uint32_t cnt = 0;
                    if (at + 4 > body.size()) return false;
                    memcpy(&cnt, body.data() + at, 4);
                    at += 4;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        uint32_t n = 0;
                        if (at + 4 > body.size()) return false;
                        memcpy(&n, body.data() + at, 4);
                        at += 4;
                        if (at + n > body.size()) return false;
                        dst.emplace_back(body.data() + at, n);
                        at += n;
                    }
                    return true;
                };
                uint32_t ncwd = 0;
                if (get(args) && get(env) && at + 4 <= body.size() && (memcpy(&ncwd, body.data() + at, 4), at + 4 + ncwd <= body.size()) &&
                    chdir(string(body.data() + at + 4, ncwd).c_str()) == 0) {
                    for (auto& nm : applied) unsetenv(nm.c_str());
                    applied.clear();
                    for (auto& kv : env) {
                        const size_t eq = kv.find('=');
                        if (eq == string::npos) continue;
                        setenv(kv.substr(0, eq).c_str(), kv.c_str() + eq + 1, 1);
                        applied.push_back(kv.substr(0, eq));
                    }
                    // the job writes where its caller would have
                    fflush(nullptr);
                    const int keep1 = dup(1), keep2 = dup(2);
                    dup2(fds[0], 1);
                    dup2(fds[1], 2);
                    std::vector<char*> av;
                    for (auto& a : args) av.push_back(&a[0]);
                    av.push_back(nullptr);
                    optind = 0;  // (glibc: scan from the start again)
                    VERBOSE = false;
                    try {
                        rc = main_cluster(int(args.size()), av.data());
                    } catch (const JobExit& e) {
                        rc = e.rc;
                    } catch (const std::exception& e) {
                        std::cerr << "isONclust2-hip: " << e.what() << std::endl;
                        rc = 1;
                    }
                    std::cout.flush();
                    std::cerr.flush();
                    fflush(nullptr);
                    dup2(keep1, 1);
                    dup2(keep2, 2);
                    close(keep1);
                    close(keep2);
                    (void)!chdir("/");  // (the caller's directory is the caller's again)
                }
            }
        }
        if (fds[0] >= 0) close(fds[0]);
        if (fds[1] >= 0) close(fds[1]);
        const int32_t out = rc;
        (void)write_all(cs, &out, 4);
        close(cs);
        if (quit) break;
    }
    (void)unlink(sock.c_str());
    close(ls);
    fflush(nullptr);
    _exit(0);  // (as the one-shot command: the driver takes the context back)
}

static int connect_to(const string& sock)
{
    const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un sa{};
    sa.sun_family = AF_UNIX;
    if (fd < 0 || sock.size() >= sizeof(sa.sun_path)) {
        if (fd >= 0) close(fd);
        return -1;
    }
    memcpy(sa.sun_path, sock.c_str(), sock.size() + 1);
    if (connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0) {
        close(fd);
        return -1;
    }
    return fd;
}

static bool worker_alive(const string& base)
{
    const int lk = open((base + ".worker").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (lk < 0) return false;
    const bool alive = flock(lk, LOCK_EX | LOCK_NB) != 0;
    close(lk);  // (if we got the lock this gives it back)
    return alive;
}

static bool spawn_worker(int slot)
{
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    if (n <= 0) return false;
    exe[n] = 0;
    const pid_t pid = fork();  // (nothing in this process has touched the GPU: the HIP runtime starts lazily)
    if (pid < 0) return false;
    if (pid == 0) {
        if (fork() != 0) _exit(0);  // the worker is nobody's child
        setsid();
        const int dn = open("/dev/null", O_RDWR);
        const string logp = slot_base(slot) + ".log";
        const int lg = getenv("ISONCLUST2_SERVE_LOG") ? open(logp.c_str(), O_CREAT | O_WRONLY | O_APPEND, 0600) : -1;
        dup2(dn, 0);
        dup2(lg >= 0 ? lg : dn, 1);
        dup2(lg >= 0 ? lg : dn, 2);
        for (int fd = 3; fd < 256; ++fd) close(fd);
        const string sl = std::to_string(slot);
        execl(exe, exe, "serve", "worker", sl.c_str(), static_cast<char*>(nullptr));
        _exit(127);
    }
    int st = 0;
    (void)waitpid(pid, &st, 0);
    return true;
}

// Hands `cluster` (argv as main_cluster takes it) to a worker.  Returns the job's exit code, or -1 when no worker could be
// reached before anything was sent: the caller then runs the job itself.
static int client(int argc, char** argv)
{
    const char* on = getenv("ISONCLUST2_SERVE");
    if (on && atoi(on) == 0) return -1;
    // (the help text leaves through exit(): answered here)
    std::vector<string> args(argv, argv + argc);
    for (size_t i = 1; i < args.size(); ++i)
        if (args[i] == "-h" || args[i] == "--help") return -1;
    char cwd[4096];
    if (!getcwd(cwd, sizeof(cwd))) return -1;
    const string dir = dir_path();
    if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return -1;
    struct stat sb;
    if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & 077) != 0) return -1;
    int max_slots = 5;  // (a card takes few processes at once; each resident worker keeps its buffers)
    if (const char* e = getenv("ISONCLUST2_SERVE_SLOTS")) max_slots = std::max(1, std::min(32, atoi(e)));
    // a slot nobody is using: the caller's lock on it lasts for the job
    int slot = -1, lk = -1;
    for (int pass = 0; pass < 2 && slot < 0; ++pass)
        for (int i = 0; i < max_slots; ++i) {
            const int fd = open((slot_base(i) + ".lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
            if (fd < 0) return -1;
            // first pass: a free slot whose worker is warm; second pass: any free slot; failing that, wait for slot 0
            if (flock(fd, LOCK_EX | LOCK_NB) == 0 && (pass == 1 || worker_alive(slot_base(i)))) {
                slot = i;
                lk = fd;
                break;
            }
            close(fd);
        }
    if (slot < 0) {
        lk = open((slot_base(0) + ".lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
        if (lk < 0 || flock(lk, LOCK_EX) != 0) return -1;
        slot = 0;
    }
    const string base = slot_base(slot), sock = base + ".sock";
    if (sock.size() >= sizeof(sockaddr_un{}.sun_path)) {  // (a directory too deep for a socket's name: the job runs here)
        close(lk);
        return -1;
    }
    int cs = worker_alive(base) ? connect_to(sock) : -1;
    if (cs < 0) {
        if (!spawn_worker(slot)) {
            close(lk);
            return -1;
        }
        for (int tries = 0; tries < 5000 && cs < 0; ++tries) {  // (the worker binds its socket before it does anything else)
            cs = connect_to(sock);
            if (cs < 0) usleep(1000);
            if (cs < 0 && tries > 200 && !worker_alive(base)) break;  // (it could not start: no point in waiting)
        }
        if (cs < 0) {
            close(lk);
            return -1;
        }
    }
    string body;
    {
        const uint32_t na = uint32_t(args.size());
        body.append(reinterpret_cast<const char*>(&na), 4);
        for (auto& a : args) put_str(body, a);
        std::vector<string> env;
        for (char** e = environ; e && *e; ++e)
            if (strncmp(*e, "IOC_", 4) == 0 || strncmp(*e, "ISONCLUST2_", 11) == 0) env.emplace_back(*e);
        const uint32_t ne = uint32_t(env.size());
        body.append(reinterpret_cast<const char*>(&ne), 4);
        for (auto& e : env) put_str(body, e);
        put_str(body, cwd);  // relative paths mean what they mean to the caller
    }
    uint32_t hdr[2] = {MAGIC, uint32_t(body.size())};
    int fds[2] = {1, 2};
    iovec io{hdr, sizeof(hdr)};
    alignas(cmsghdr) char cbuf[CMSG_SPACE(sizeof(fds))];
    memset(cbuf, 0, sizeof(cbuf));
    msghdr mh{};
    mh.msg_iov = &io;
    mh.msg_iovlen = 1;
    mh.msg_control = cbuf;
    mh.msg_controllen = sizeof(cbuf);
    cmsghdr* cm = CMSG_FIRSTHDR(&mh);
    cm->cmsg_level = SOL_SOCKET;
    cm->cmsg_type = SCM_RIGHTS;
    cm->cmsg_len = CMSG_LEN(sizeof(fds));
    memcpy(CMSG_DATA(cm), fds, sizeof(fds));
    ssize_t w;
    do w = sendmsg(cs, &mh, MSG_NOSIGNAL); while (w < 0 && errno == EINTR);
    if (w != ssize_t(sizeof(hdr))) {
        close(cs);
        close(lk);
        return -1;
    }
    int32_t rc = 1;
    if (!write_all(cs, body.data(), body.size()) || !read_all(cs, &rc, 4)) {
        // the job was handed over and the worker is gone (a crash takes its context with it): say so, do not run it twice
        std::cerr << "isONclust2-hip: the resident worker ended before the job did (ISONCLUST2_SERVE=0 runs the job in the calling process)" << std::endl;
        rc = 1;
    }
    close(cs);
    close(lk);
    return rc;
}

static int stop_all()
{
    int stopped = 0;
    for (int i = 0; i < 32; ++i) {
        const string base = slot_base(i);
        if (!worker_alive(base)) continue;
        const int cs = connect_to(base + ".sock");
        if (cs < 0) continue;
        uint32_t hdr[2] = {MAGIC, 0xFFFFFFFFu};
        int32_t rc = 0;
        if (write_all(cs, hdr, sizeof(hdr)) && read_all(cs, &rc, 4)) ++stopped;
        close(cs);
        for (int t = 0; t < 2000 && worker_alive(base); ++t) usleep(1000);  // (its context is gone when this returns)
    }
    std::cerr << "isONclust2-hip: " << stopped << " worker(s) stopped" << std::endl;
    return 0;
}

}  // namespace serve

int main(int argc, char** argv)
{
    if (argc < 2) {
        cerr << "isONclust2-hip <sort|cluster|dump|info|version|help> ..." << endl;
        return 0;
    }
    const string cmd = argv[1];
    // The GPU commands leave through _exit once their files are closed and the streams flushed: the HIP runtime's exit handlers
    // (code objects unloaded, every allocation returned one by one) cost a one-shot process 50 - 100 ms for nothing — the driver
    // takes the process's memory back either way.  IOC_CLI_CLEAN_EXIT=1: the ordinary way out (profilers and sanitizers write their results from exit handlers).
    auto leave = [](int rc) {
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        if (getenv("IOC_CLI_CLEAN_EXIT")) return rc;
        _exit(rc);
    };
    if (cmd == "sort") return leave(main_sort(argc - 1, argv + 1));
    if (cmd == "cluster") {
        const int served = serve::client(argc - 1, argv + 1);
        if (served >= 0) return served;
        return leave(main_cluster(argc - 1, argv + 1));
    }
    if (cmd == "serve") {
        const string what = argc > 2 ? argv[2] : "";
        if (what == "worker" && argc > 3) return serve::worker(atoi(argv[3]));
        if (what == "stop") return serve::stop_all();
        std::cerr << "isONclust2-hip serve stop    (workers are started by `cluster` itself; ISONCLUST2_SERVE=0 turns them off)" << endl;
        return 0;
    }
    if (cmd == "dump") return main_dump(argc - 1, argv + 1);
    if (cmd == "info") return main_info(argc - 1, argv + 1);
    if (cmd == "selftest") return main_selftest(argc - 1, argv + 1);
    if (cmd == "golden") {  // writes the golden batch of the byte-layout tests (host only)
        if (argc < 3) die("isONclust2-hip golden out.cer");
        Batch g;
        make_golden_batch(g);
        string err;
        if (!save_batch(g, argv[2], err)) die(err);
        return 0;
    }
    if (cmd == "version") {
        cerr << "isONclust2 version: " << VERSION << endl;
        return 0;
    }
    if (cmd == "help") {
        cerr << "isONclust2-hip: sort, cluster, dump, info, version, help — flags as in isONclust2 v2.4" << endl;
        return 0;
    }
    cerr << "Invalid command: " << cmd << endl;
    return 0;
}
