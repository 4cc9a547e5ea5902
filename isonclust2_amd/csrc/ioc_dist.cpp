// ioc_dist.cpp — the multi-GPU exchange step of the path in C++ over RCCL (BASELINE.json north_star: "host
// code stays C++", "RCCL all-gather over xGMI of cluster representatives at merge").  One process and one context per GPU;
// a context owns ONE communicator.  Initial clustering needs no communication (the reference's pipeline runs one `cluster`
// process per batch, README.md:105-117); the merge (`cluster -l L -r R`, src/cluster.cpp:67-322 with two batches) matches
// every right cluster through its representative's Mins / RevMins (src/cluster.cpp:537-539), so what travels is the
// representatives' records: their minimizer lists HBM to HBM (ragged: one broadcast per rank inside one RCCL group, straight
// into the [all forward lists][all reverse lists] layout ioc_batch_view takes — no padding, no re-layout copy), and a small
// per-representative host record (lengths, error rates, raw sequences in sahlin / furious mode) staged through HBM.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // (types and prototypes only: the library itself is opened on first use, see rccl())

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ioc_internal.h"

// librccl.so is 573 MB of device code for every collective, data type and architecture: linked the ordinary way it is mapped,
// relocated and registered with the HIP runtime at EVERY start of every process that loads this library — the one-GPU `cluster`
// command included, which never communicates (0.1 - 0.2 s of a 0.5 s process, round 3's VERDICT item 7).  So it is opened when a
// context first asks for a communicator, and the ten entry points used here are looked up by name.
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
// the library, opened once per process (thread-safe: a function-local static); nullptr + RcclApi::error if it cannot be had
const RcclApi* rccl(std::string* why = nullptr)
{
    static const RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
        }
        if (!a.handle) {
            const char* e = dlerror();
            a.error = std::string("librccl.so.1 cannot be opened: ") + (e ? e : "?");
            return a;
        }
        bool ok = true;
        auto sym = [&](const char* nm) {
            void* p = dlsym(a.handle, nm);
            if (!p) {
                ok = false;
                a.error = std::string("librccl.so.1 lacks ") + nm;
            }
            return p;
        };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) a.handle = nullptr;
        return a;
    }();
    if (!api.handle) {
        if (why) *why = api.error;
        return nullptr;
    }
    return &api;
}
}  // namespace

struct ioc_dist_state {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf stage;  // host records on their way through HBM
    // status agreement (see agree()): a device word pair and its pinned mirror; `informed` = this rank has already learnt
    // through an agreement that some rank failed, so it owes its peers no further announcement
    int32_t* d_status = nullptr;
    int32_t* h_status = nullptr;
    bool informed = false;
    hipStream_t stream = nullptr;
};

#define NCK(c, x)                                                                                   \
    do {                                                                                            \
        ncclResult_t r_ = (x);                                                                      \
        if (r_ != ncclSuccess) {                                                                    \
            dist_abort((c)->dist);                                                                  \
            return ioc_fail(c, IOC_ERR_HIP, std::string("RCCL: ") + rccl()->GetErrorString(r_) + " (" #x "); communicator aborted"); \
        }                                                                                           \
    } while (0)
#define HCK(c, x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) return ioc_fail(c, IOC_ERR_HIP, std::string(hipGetErrorString(e_)) + " (" #x ")"); \
    } while (0)

static int need_dist(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    if (!c->dist) return ioc_fail(c, IOC_ERR_STATE, "ioc_dist_init has not been called on this context");
    if (!c->dist->comm) return ioc_fail(c, IOC_ERR_STATE, "the communicator of this context was aborted after an RCCL error");
    return IOC_OK;
}

// An RCCL call failed on this rank: abort the communicator, so that this rank's queued collectives are torn down and every
// later ioc_dist_* call fails at once instead of entering a collective the peers will never complete.
static void dist_abort(ioc_dist_state* d)
{
    if (d && d->comm) {
        (void)rccl()->CommAbort(d->comm);
        d->comm = nullptr;
    }
}

// STATUS AGREEMENT.  A rank that fails locally (an allocation, a size check, a replay that finds nothing) must not simply
// return: its peers are inside, or about to enter, the next collective and would wait for ever.  Protocol: every collective
// stage that can be preceded by a local failure starts with agree(local status) — one all-reduce (max) of one word, read back
// by the host.  All ranks leave it with the same answer; when it is non-zero every rank gives up at that point.  A rank that
// fails BETWEEN two agreements jumps straight to its next agree() with status 1 — that call pairs with whatever agreement the
// peers reach next (they are all the same collective: 1 x int32, max), the peers learn of the failure there and stop.  A rank
// that has been told of a failure this way (`informed`) skips its own closing announcement: everybody has left already.
static int agree(ioc_dist_state* d, int local_status)
{
    if (!d || !d->comm) return 1;
    if (d->informed) return 1;
    if (d->world == 1) return local_status != 0;
    d->h_status[0] = local_status != 0 ? 1 : 0;
    if (hipMemcpyAsync(d->d_status, d->h_status, 4, hipMemcpyHostToDevice, d->stream) != hipSuccess ||
        rccl()->AllReduce(d->d_status, d->d_status + 1, 1, ncclInt32, ncclMax, d->comm, d->stream) != ncclSuccess) {
        dist_abort(d);
        return 1;
    }
    if (hipMemcpyAsync(d->h_status + 1, d->d_status + 1, 4, hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
        hipStreamSynchronize(d->stream) != hipSuccess) {
        dist_abort(d);
        return 1;
    }
    const int worst = static_cast<volatile int32_t*>(d->h_status)[1];
    if (worst != 0 && local_status == 0) d->informed = true;  // (somebody else failed: nothing left to announce)
    return worst;
}

static int reserve(ioc_ctx* c, DevBuf& b, size_t bytes)
{
    if (b.cap >= bytes && b.p) return IOC_OK;
    if (b.p) {
        HCK(c, hipStreamSynchronize(c->stream));
        HCK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HCK(c, hipMalloc(&b.p, bytes ? bytes : 256));
    b.cap = bytes ? bytes : 256;
    return IOC_OK;
}

// ioc_set_shard's exchange over the context's communicator: in-place all-reduce on the context's stream
static int rccl_exchange(void* user, void* d_buf, int64_t count, int32_t kind, void* hip_stream)
{
    ioc_dist_state* d = static_cast<ioc_dist_state*>(user);
    if (!d || !d->comm || count < 0) return 1;
    // (every exchange of the sharded resolve is a point where the ranks agree that all of them are still in: a rank that
    // failed since the last one announces it here, through ioc_dist_merge's closing agree())
    if (agree(d, 0) != 0) return 1;
    if (count == 0) return 0;
    ncclDataType_t t = kind == IOC_XCHG_MAX_U8 ? ncclUint8 : kind == IOC_XCHG_MIN_U32 ? ncclUint32 : ncclInt32;
    ncclRedOp_t op = kind == IOC_XCHG_MAX_U8 ? ncclMax : kind == IOC_XCHG_MIN_U32 ? ncclMin : ncclSum;
    if (rccl()->AllReduce(d_buf, d_buf, size_t(count), t, op, d->comm, static_cast<hipStream_t>(hip_stream)) != ncclSuccess) {
        dist_abort(d);
        return 1;
    }
    return 0;
}

extern "C" {

int ioc_dist_unique_id(uint8_t* id)
{
    if (!id) return IOC_ERR_ARG;
    static_assert(IOC_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    const RcclApi* R = rccl();
    if (!R || R->GetUniqueId(&u) != ncclSuccess) return IOC_ERR_HIP;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return IOC_OK;
}

int ioc_dist_init(ioc_ctx* c, const uint8_t* id, int32_t rank, int32_t world)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return IOC_ERR_ARG;
    if (c->dist) return ioc_fail(c, IOC_ERR_STATE, "ioc_dist_init: the context already has a communicator");
    HCK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ioc_dist_state* d = new ioc_dist_state;
    d->rank = rank;
    d->world = world;
    std::string why;
    const RcclApi* R = rccl(&why);
    if (!R) {
        delete d;
        return ioc_fail(c, IOC_ERR_HIP, "RCCL: " + why);
    }
    ncclResult_t r = R->CommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete d;
        return ioc_fail(c, IOC_ERR_HIP, std::string("RCCL: ncclCommInitRank: ") + rccl()->GetErrorString(r));
    }
    d->stream = c->stream;
    if (hipMalloc(reinterpret_cast<void**>(&d->d_status), 16) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&d->h_status), 16, hipHostMallocDefault) != hipSuccess) {
        if (d->d_status) (void)hipFree(d->d_status);
        (void)rccl()->CommAbort(d->comm);
        delete d;
        return ioc_fail(c, IOC_ERR_HIP, "ioc_dist_init: no memory for the status words");
    }
    c->dist = d;
    return IOC_OK;
}

int ioc_dist_shutdown(ioc_ctx* c)
{
    if (!c) return IOC_ERR_ARG;
    if (!c->dist) return IOC_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->dist->comm) (void)rccl()->CommDestroy(c->dist->comm);
    if (c->dist->stage.p) (void)hipFree(c->dist->stage.p);
    if (c->dist->d_status) (void)hipFree(c->dist->d_status);
    if (c->dist->h_status) (void)hipHostFree(c->dist->h_status);
    delete c->dist;
    c->dist = nullptr;
    return IOC_OK;
}

int ioc_dist_info(const ioc_ctx* c, int32_t* rank, int32_t* world)
{
    if (!c) return IOC_ERR_ARG;
    if (rank) *rank = c->dist ? c->dist->rank : 0;
    if (world) *world = c->dist ? c->dist->world : 1;
    return IOC_OK;
}

// every rank's `bytes` bytes (the same number on every rank) from device memory into [world][bytes] on every rank
int ioc_dist_allgather_device(ioc_ctx* c, const void* d_send, void* d_recv, int64_t bytes)
{
    if (int rc = need_dist(c)) return rc;
    if (bytes < 0 || (bytes && (!d_send || !d_recv))) return IOC_ERR_ARG;
    HCK(c, hipSetDevice(c->device));
    if (bytes) NCK(c, rccl()->AllGather(d_send, d_recv, size_t(bytes), ncclUint8, c->dist->comm, c->stream));
    return IOC_OK;
}

// ragged: rank r contributes counts[r] elements of `esize` bytes; block r lands at d_recv + displs[r] * esize on every rank
// (one broadcast per rank inside one RCCL group).  counts / displs are host arrays every rank passes identically.
int ioc_dist_allgatherv_device(ioc_ctx* c, const void* d_send, void* d_recv, const int64_t* counts, const int64_t* displs,
                               int32_t esize)
{
    if (int rc = need_dist(c)) return rc;
    if (!counts || !displs || esize <= 0) return IOC_ERR_ARG;
    HCK(c, hipSetDevice(c->device));
    ioc_dist_state* d = c->dist;
    NCK(c, rccl()->GroupStart());
    for (int r = 0; r < d->world; ++r) {
        if (counts[r] <= 0) continue;
        char* dst = static_cast<char*>(d_recv) + displs[r] * esize;
        // (the root's send buffer may be its own slot of d_recv: in place)
        ncclResult_t e = rccl()->Broadcast(r == d->rank ? d_send : dst, dst, size_t(counts[r]) * size_t(esize), ncclUint8, r, d->comm, c->stream);
        if (e != ncclSuccess) {
            (void)rccl()->GroupEnd();
            const std::string why = std::string("RCCL: ncclBroadcast: ") + rccl()->GetErrorString(e);
            dist_abort(d);  // (as NCK does: later calls must fail at once instead of entering collectives on a half-used communicator)
            return ioc_fail(c, IOC_ERR_HIP, why);
        }
    }
    NCK(c, rccl()->GroupEnd());
    return IOC_OK;
}

int ioc_dist_allgather_i64(ioc_ctx* c, int64_t mine, int64_t* all)
{
    if (int rc = need_dist(c)) return rc;
    if (!all) return IOC_ERR_ARG;
    ioc_dist_state* d = c->dist;
    HCK(c, hipSetDevice(c->device));
    if (int rc = reserve(c, d->stage, size_t(d->world + 1) * 8)) return rc;
    int64_t* dv = static_cast<int64_t*>(d->stage.p);
    HCK(c, hipMemcpyAsync(dv + d->world, &mine, 8, hipMemcpyHostToDevice, c->stream));
    NCK(c, rccl()->AllGather(dv + d->world, dv, 1, ncclInt64, d->comm, c->stream));
    HCK(c, hipMemcpyAsync(all, dv, size_t(d->world) * 8, hipMemcpyDeviceToHost, c->stream));
    HCK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

// host records, ragged: `bytes` of this rank -> recv (sum of all ranks' bytes, rank order); sizes[world] receives the sizes.
// Call with recv == NULL first to learn the sizes.
int ioc_dist_allgatherv_host(ioc_ctx* c, const void* send, int64_t bytes, void* recv, int64_t* sizes)
{
    if (int rc = need_dist(c)) return rc;
    if (bytes < 0 || !sizes || (bytes && !send)) return IOC_ERR_ARG;
    ioc_dist_state* d = c->dist;
    if (int rc = ioc_dist_allgather_i64(c, bytes, sizes)) return rc;
    if (!recv) return IOC_OK;
    std::vector<int64_t> displs(size_t(d->world) + 1, 0);
    for (int r = 0; r < d->world; ++r) displs[size_t(r) + 1] = displs[size_t(r)] + sizes[r];
    const int64_t total = displs[size_t(d->world)];
    if (total == 0) return IOC_OK;
    if (int rc = reserve(c, d->stage, size_t(total) + 256)) return rc;
    char* dv = static_cast<char*>(d->stage.p);
    if (bytes) HCK(c, hipMemcpyAsync(dv + displs[size_t(d->rank)], send, size_t(bytes), hipMemcpyHostToDevice, c->stream));
    if (int rc = ioc_dist_allgatherv_device(c, dv + displs[size_t(d->rank)], dv, sizes, displs.data(), 1)) return rc;
    HCK(c, hipMemcpyAsync(recv, dv, size_t(total), hipMemcpyDeviceToHost, c->stream));
    HCK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_dist_allreduce_max(ioc_ctx* c, double* x)
{
    if (int rc = need_dist(c)) return rc;
    if (!x) return IOC_ERR_ARG;
    ioc_dist_state* d = c->dist;
    HCK(c, hipSetDevice(c->device));
    if (int rc = reserve(c, d->stage, 16)) return rc;
    double* dv = static_cast<double*>(d->stage.p);
    HCK(c, hipMemcpyAsync(dv, x, 8, hipMemcpyHostToDevice, c->stream));
    NCK(c, rccl()->AllReduce(dv, dv + 1, 1, ncclDouble, ncclMax, d->comm, c->stream));
    HCK(c, hipMemcpyAsync(x, dv + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HCK(c, hipStreamSynchronize(c->stream));
    return IOC_OK;
}

int ioc_dist_barrier(ioc_ctx* c)
{
    double x = 0;
    return ioc_dist_allreduce_max(c, &x);
}

// ---- the merge of all ranks' freshly clustered batches as ONE pass on every rank ---------------------------------------
// per-representative host record that travels: 8 x u32/f64 fields + the representative's raw sequence (sahlin / furious)
struct RepMeta {
    uint32_t raw_len, hpc_len, n_fwd, n_rev;
    uint32_t state, pad;
    double score, raw_err, hpc_err;
};

// the all-reduce ioc_dist_merge hands to ioc_set_shard, callable by a host program that drives ioc_cluster_merge itself
int ioc_dist_exchange(ioc_ctx* c, void* d_buf, int64_t count, int32_t kind)
{
    if (int rc = need_dist(c)) return rc;
    if (!d_buf || count < 0 || kind < IOC_XCHG_MAX_U8 || kind > IOC_XCHG_SUM_I32) return IOC_ERR_ARG;
    HCK(c, hipSetDevice(c->device));
    if (rccl_exchange(c->dist, d_buf, count, kind, c->stream) != 0) return ioc_fail(c, IOC_ERR_HIP, "RCCL: ncclAllReduce failed");
    return IOC_OK;
}

int ioc_dist_set_shard(ioc_ctx* c, int32_t on)
{
    if (int rc = need_dist(c)) return rc;
    if (on && c->dist->world > 1) {
        c->dist->informed = false;  // (a failure a peer announced during an earlier sharded call is that call's: ADVICE r4)
        return ioc_set_shard(c, c->dist->world, c->dist->rank, rccl_exchange, c->dist);
    }
    return ioc_set_shard(c, 1, 0, nullptr, nullptr);
}

int ioc_dist_merge(ioc_ctx* c, const ioc_params* p, const char* table_path, const ioc_batch_view* reps, int32_t min_cls_size,
                   int64_t out_cap, int32_t* out_cls, int8_t* out_strand, int64_t* out_counts, ioc_cluster_stats* stats,
                   ioc_dist_merge_times* times)
{
    if (int rc = need_dist(c)) return rc;
    if (!p || !reps || !out_counts || reps->n < 0) return IOC_ERR_ARG;
    ioc_dist_state* d = c->dist;
    const int W = d->world;
    HCK(c, hipSetDevice(c->device));
    hipEvent_t e0, e1;
    HCK(c, hipEventCreate(&e0));
    HCK(c, hipEventCreate(&e1));
    struct EvGuard {
        hipEvent_t a, b;
        ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    } evg{e0, e1};
    const int32_t n = reps->n;
    const int64_t f0 = n ? reps->off_fwd[0] : 0, fw = n ? reps->off_fwd[n] - f0 : 0;
    const int64_t r0 = n ? reps->off_rev[0] : 0, rw = n ? reps->off_rev[n] - r0 : 0;
    const bool have_seq = reps->raw_seq && reps->raw_off;
    const int64_t sb = have_seq && n ? reps->raw_off[n] - reps->raw_off[0] : 0;
    // 1. sizes: clusters, forward words, reverse words, sequence bytes (-1: this rank has no sequences) of every rank
    std::vector<int64_t> cnt(static_cast<size_t>(W)), fws(static_cast<size_t>(W)), rws(static_cast<size_t>(W)), sbs(static_cast<size_t>(W));
    if (int rc = ioc_dist_allgather_i64(c, n, cnt.data())) return rc;
    if (int rc = ioc_dist_allgather_i64(c, fw, fws.data())) return rc;
    if (int rc = ioc_dist_allgather_i64(c, rw, rws.data())) return rc;
    if (int rc = ioc_dist_allgather_i64(c, have_seq ? sb : -1, sbs.data())) return rc;
    int64_t N = 0, FW = 0, RW = 0, SB = 0;
    bool all_seq = true;
    for (int r = 0; r < W; ++r) {
        out_counts[r] = cnt[size_t(r)];
        N += cnt[size_t(r)];
        FW += fws[size_t(r)];
        RW += rws[size_t(r)];
        if (sbs[size_t(r)] < 0) all_seq = false; else SB += sbs[size_t(r)];
    }
    if (!out_cls || !out_strand) return IOC_OK;  // sizing call
    d->informed = false;
    // (local checks first, then the ranks agree: a rank that cannot go on must not leave the others inside the list exchange)
    int local = IOC_OK;
    if (out_cap < N) local = ioc_fail(c, IOC_ERR_ARG, "ioc_dist_merge: out_cap is smaller than the number of representatives of all ranks");
    else if (N > INT32_MAX) local = ioc_fail(c, IOC_ERR_CAPACITY, "too many representatives");
    // 2. the minimizer lists, HBM to HBM, into [all forward lists][all reverse lists]
    if (local == IOC_OK) local = reserve(c, c->b_dist_min, size_t(FW + RW) * 4 + 256);
    if (local == IOC_OK) local = reserve(c, c->b_dist_pos, size_t(FW + RW) * 4 + 256);
    if (agree(d, local) != 0)
        return local != IOC_OK ? local : ioc_fail(c, IOC_ERR_STATE, "ioc_dist_merge: another rank could not start the exchange (its own error says why)");
    // Everything between the opening agreement and the closing one runs inside `exchange_and_merge`: whatever way it ends — a HIP call
    // that fails, a host allocation, an inconsistent record — this rank still reaches the closing agreement and says so there, where
    // the peers' next agreement meets it (the early returns of this stretch used to leave them waiting).
    float ms_comm = 0;
    std::vector<RepMeta> mine(static_cast<size_t>(n));
    int exchanges = 0;
    bool shard = false;
    auto t0 = std::chrono::steady_clock::now();
    auto exchange_and_merge = [&]() -> int {
        HCK(c, hipEventRecord(e0, c->stream));
        uint32_t* gmin = static_cast<uint32_t*>(c->b_dist_min.p);
        uint32_t* gpos = static_cast<uint32_t*>(c->b_dist_pos.p);
        std::vector<int64_t> dF(static_cast<size_t>(W)), dR(static_cast<size_t>(W));
        {
            int64_t a = 0, b = FW;
            for (int r = 0; r < W; ++r) {
                dF[size_t(r)] = a;
                dR[size_t(r)] = b;
                a += fws[size_t(r)];
                b += rws[size_t(r)];
            }
        }
        const hipMemcpyKind kind = reps->minimizers_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (fw) {
            HCK(c, hipMemcpyAsync(gmin + dF[size_t(d->rank)], reps->min_val + f0, size_t(fw) * 4, kind, c->stream));
            HCK(c, hipMemcpyAsync(gpos + dF[size_t(d->rank)], reps->min_pos + f0, size_t(fw) * 4, kind, c->stream));
        }
        if (rw) {
            HCK(c, hipMemcpyAsync(gmin + dR[size_t(d->rank)], reps->min_val + r0, size_t(rw) * 4, kind, c->stream));
            HCK(c, hipMemcpyAsync(gpos + dR[size_t(d->rank)], reps->min_pos + r0, size_t(rw) * 4, kind, c->stream));
        }
        if (int rc = ioc_dist_allgatherv_device(c, gmin + dF[size_t(d->rank)], gmin, fws.data(), dF.data(), 4)) return rc;
        if (int rc = ioc_dist_allgatherv_device(c, gmin + dR[size_t(d->rank)], gmin, rws.data(), dR.data(), 4)) return rc;
        if (int rc = ioc_dist_allgatherv_device(c, gpos + dF[size_t(d->rank)], gpos, fws.data(), dF.data(), 4)) return rc;
        if (int rc = ioc_dist_allgatherv_device(c, gpos + dR[size_t(d->rank)], gpos, rws.data(), dR.data(), 4)) return rc;
        HCK(c, hipEventRecord(e1, c->stream));
        // 3. the per-representative host records (and the raw sequences)
        std::vector<RepMeta> all(static_cast<size_t>(N));
        for (int32_t i = 0; i < n; ++i) {
            RepMeta& m = mine[size_t(i)];
            m.raw_len = reps->raw_len[i];
            m.hpc_len = reps->hpc_len[i];
            m.n_fwd = uint32_t(reps->off_fwd[i + 1] - reps->off_fwd[i]);
            m.n_rev = uint32_t(reps->off_rev[i + 1] - reps->off_rev[i]);
            m.state = reps->state[i];
            m.pad = 0;
            m.score = reps->score[i];
            m.raw_err = reps->raw_err[i];
            m.hpc_err = reps->hpc_err[i];
        }
        std::vector<int64_t> sz(static_cast<size_t>(W));
        if (int rc = ioc_dist_allgatherv_host(c, mine.data(), int64_t(mine.size() * sizeof(RepMeta)), all.data(), sz.data())) return rc;
        std::string seq_all;
        std::vector<int64_t> seq_off;
        if (all_seq) {
            seq_all.resize(size_t(SB));
            std::vector<int64_t> ssz(static_cast<size_t>(W));
            if (int rc = ioc_dist_allgatherv_host(c, have_seq && n ? reps->raw_seq + reps->raw_off[0] : nullptr, sb, seq_all.data(), ssz.data()))
                return rc;
            // the sequence lengths are raw_len (RawSeq->Str().length())
            seq_off.resize(size_t(N) + 1);
            seq_off[0] = 0;
            for (int64_t i = 0; i < N; ++i) seq_off[size_t(i) + 1] = seq_off[size_t(i)] + all[size_t(i)].raw_len;
            if (seq_off[size_t(N)] != SB) return ioc_fail(c, IOC_ERR_INPUT, "ioc_dist_merge: raw_len does not add up to the sequences gathered");
        }
        HCK(c, hipEventSynchronize(e1));
        (void)hipEventElapsedTime(&ms_comm, e0, e1);
        // 4. the combined view: rank 0's representatives are clusters from the start (the left fold ((b0 + b1) + b2) ... of freshly
        //    clustered batches makes the decisions of one loop over all representatives in rank order: ioc_batch_view::is_cluster)
        std::vector<int64_t> off_f(static_cast<size_t>(N) + 1), off_r(static_cast<size_t>(N) + 1);
        std::vector<uint32_t> raw_len(static_cast<size_t>(N)), hpc_len(static_cast<size_t>(N));
        std::vector<double> score(static_cast<size_t>(N)), raw_err(static_cast<size_t>(N)), hpc_err(static_cast<size_t>(N));
        std::vector<uint8_t> state(static_cast<size_t>(N)), is_cluster(static_cast<size_t>(N), 0);
        off_f[0] = 0;
        off_r[0] = FW;
        for (int64_t i = 0; i < N; ++i) {
            const RepMeta& m = all[size_t(i)];
            off_f[size_t(i) + 1] = off_f[size_t(i)] + m.n_fwd;
            off_r[size_t(i) + 1] = off_r[size_t(i)] + m.n_rev;
            raw_len[size_t(i)] = m.raw_len;
            hpc_len[size_t(i)] = m.hpc_len;
            score[size_t(i)] = m.score;
            raw_err[size_t(i)] = m.raw_err;
            hpc_err[size_t(i)] = m.hpc_err;
            state[size_t(i)] = uint8_t(m.state);
            is_cluster[size_t(i)] = i < cnt[0] ? 1 : 0;
        }
        if (off_f[size_t(N)] != FW || off_r[size_t(N)] != FW + RW) return ioc_fail(c, IOC_ERR_INPUT, "ioc_dist_merge: list lengths do not add up");
        ioc_batch_view v{};
        v.n = int32_t(N);
        v.off_fwd = off_f.data();
        v.off_rev = off_r.data();
        v.min_val = gmin;
        v.min_pos = gpos;
        v.total = FW + RW;
        v.raw_len = raw_len.data();
        v.hpc_len = hpc_len.data();
        v.score = score.data();
        v.raw_err = raw_err.data();
        v.hpc_err = hpc_err.data();
        v.state = state.data();
        v.min_qual = reps->min_qual;
        v.raw_seq = all_seq ? seq_all.data() : nullptr;
        v.raw_off = all_seq ? seq_off.data() : nullptr;
        v.n_members = nullptr;
        v.depth = 0;
        v.min_cls_size = min_cls_size;
        v.is_cluster = is_cluster.data();
        v.minimizers_on_device = 1;
        HCK(c, hipStreamSynchronize(c->stream));
        t0 = std::chrono::steady_clock::now();
        // fast mode: score + resolve sharded over the ranks (query j on rank j % W), `valid` all-reduced after every sweep;
        // sahlin / furious: the alignment rounds sharded by owner of the query, their verdicts summed over the ranks (ioc_host.cpp)
        const char* es = getenv("IOC_DIST_SHARD");
        shard = W > 1 && !(es && es[0] == '0');
        int rc = IOC_OK;
        if (shard && ioc_dist_set_shard(c, 1) != IOC_OK) rc = IOC_ERR_STATE;
        if (rc == IOC_OK) rc = ioc_cluster_merge(c, p, table_path, nullptr, &v, out_cls, out_strand, stats);
        exchanges = ioc_shard_exchanges(c);
        if (shard) (void)ioc_dist_set_shard(c, 0);
        return rc;
    };
    int rc = IOC_OK;
    try {
        rc = exchange_and_merge();
    } catch (const std::exception& e) {
        rc = ioc_fail(c, IOC_ERR_STATE, std::string("ioc_dist_merge: ") + e.what());
    }
    if (shard) (void)ioc_dist_set_shard(c, 0);
    // the closing agreement: a rank that failed since the last exchange (or in a merge without exchanges: the alignment modes)
    // says so here; it pairs with the agreement in front of the peers' next exchange or with their closing one
    const bool told = d->informed;
    if (!told && agree(d, rc) != 0 && rc == IOC_OK)
        rc = ioc_fail(c, IOC_ERR_STATE, "ioc_dist_merge: another rank failed inside the merge (its own error says why)");
    if (times) {
        times->sharded = shard ? 1 : 0;
        times->exchanges = exchanges;
        times->ms_exchange_lists = ms_comm;
        times->ms_merge = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        times->bytes_lists = (fw + rw) * 8;
        times->bytes_records = int64_t(mine.size() * sizeof(RepMeta)) + (all_seq ? sb : 0);
    }
    return rc;
}

}  // extern "C"
