/*
 * ref_shim.cpp — extern "C" wrappers around the reference translation units that compile
 * stand-alone from /root/reference/src (kmer_index.cpp, util.cpp, p_emp_prob.cpp).
 * Built ONLY in the build container, into oracle/_ref/ (git-ignored); used by tests to
 * validate the oracle's restatement function by function.  No reference source is copied:
 * the reference files are compiled where they lie (see oracle/Makefile).
 *
 * The other hot-path TUs (minimizer.cpp, hpc.cpp, qualscore.cpp, cluster.cpp) include
 * tbb/, bioparser/, cereal/, spoa/, parasail and a cmake-generated config header, none of
 * which exist in this image, so they are unbuildable here and are pinned through the
 * reference's unit-test known answers instead (tests/golden/reference_kat.json).
 */
#include <cstdint>
#include <cstring>
#include <string>

#include "kmer_index.h"
#include "p_emp_prob.h"
#include "util.h"

extern "C" {

int ref_kmer_encode(const char* seq, int n, int k, uint32_t* out)
{
    auto v = KmerEncodeSeq(std::string(seq, size_t(n)), unsigned(k));
    if (out) memcpy(out, v.data(), v.size() * sizeof(unsigned));
    return int(v.size());
}

uint32_t ref_kmer_to_index(const char* kmer, int k)
{
    std::string s(kmer, size_t(k));
    return KmerToIndex(s, s.end());
}

void ref_index_to_kmer(uint32_t idx, int k, char* out)
{
    auto s = IndexToKmer(idx, unsigned(k));
    memcpy(out, s.data(), s.size());
}

int ref_revcomp(const char* seq, int n, char* out)
{
    try {
        auto s = RevComp(std::string(seq, size_t(n)));
        memcpy(out, s.data(), s.size());
        return 0;
    } catch (...) {
        return -1;
    }
}

double ref_round(double x, int precision) { return round(x, precision); }

void* ref_pmin_init(int k, int w) { return new MinSharedMap(InitMinSharedMap(k, w)); }
int ref_pmin_size(void* h) { return int(static_cast<MinSharedMap*>(h)->size()); }
void ref_pmin_free(void* h) { delete static_cast<MinSharedMap*>(h); }
double ref_pmin_lookup(void* h, double e1, double e2, int* err)
{
    try {
        *err = 0;
        return GetPMinShared(e1, e2, *static_cast<MinSharedMap*>(h));
    } catch (...) {
        *err = -1;
        return -1.0;
    }
}
}
