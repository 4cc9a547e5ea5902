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
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace cer {

struct Minimizer {
    uint32_t Min, Pos, Index;
};

struct Seq {  // src/seq.h:20-98
    std::string name, seq, qual;
    double score = 0, errorRate = 0;
};

struct ProcSeq {  // src/cluster_data.h:14-26
    std::unique_ptr<Seq> RawSeq, HpcSeq;
    std::vector<Minimizer> Mins, RevMins;
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

typedef std::vector<std::pair<uint32_t, std::vector<uint32_t>>> MinDB;  // kept sorted by key in memory

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

bool save_batch(const Batch& b, const std::string& path, std::string& err);
bool load_batch(Batch& b, const std::string& path, std::string& err);
bool save_sorted_idx(const std::string& fastq_path, const std::string& path);  // SortedIdx, src/output.h:15-23
bool load_sorted_idx(std::string& fastq_path, const std::string& path);

}  // namespace cer
#endif
