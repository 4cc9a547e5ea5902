"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Initial clustering shards batches one per GPU and needs no communication (the reference pipeline
runs them as separate processes, README.md:105-117).  The one exchange step of the path is the
merge (`cluster -l L -r R`, src/cluster.cpp:67-322 with two batches):

  1. every rank gathers the minimizer lists of its cluster representatives ON THE DEVICE, out of the batch's
     query arrays that are still in HBM, into two u32 device buffers (ioc_gather_records_device): the bulk of a
     representative record (~64 KB each at 16.7 kb reads) never visits the host;
  2. ragged all-gather of those two buffers as device tensors (RCCL over xGMI, HBM to HBM; sizes first), and of
     one small packed u32 record per rank with what is left: lengths, error rates, membership of the reads,
     raw sequences of the representatives in sahlin / furious mode (`pack_clustered(..., with_minimizers=False)`);
  3. every rank replays the reference's left fold ((b0 + b1) + b2) ... on its own GPU with the gathered
     buffers used in place (ioc_batch_view::minimizers_on_device) — `merge_all_device`.
  (`merge_all` is the same merge from host arrays: the path of the tests that have no second GPU.)

Step 3 is ONE pass of the merge path, not world-1 of them: with consensus off a merge only appends the right
batch's unmatched clusters to the left and extends the MinDB by their minimizers (cluster.cpp:178-217), so
folding b1, b2, ... one after the other makes exactly the decisions of one greedy loop over the representatives
of b0, b1, b2, ... in that order in which b0's are clusters from the start (ioc_batch_view::is_cluster; same order
of queries, same left state at every query, `right.Depth == 0` for freshly clustered batches so the MinClsSize gate
of cluster.cpp:119-123 is off; the left MinDB is what AddMinimizers makes of b0's representatives, so it need not
travel) — checked against the oracle's step-by-step fold in tests/test_gpu_merge.py and tests/test_gpu_fullsize.py.
All-pairs scoring + fixed-point resolve of that one pass run on every rank (replicated: the result is needed
everywhere, and a rank's share of the candidate tables would be ~N^2 x 12 B to exchange — two orders of magnitude
more than the representative records themselves).
"""
import os
import time

import numpy as np

from .pipeline import VIEW_KEYS, ClusteredBatch, cluster_merge, concat_records, gather_records, gather_seqs

_MAGIC = 0x494F4332  # "IOC2"
_U32_FIELDS = ("raw_len", "hpc_len", "state")
_F64_FIELDS = ("score", "raw_err", "hpc_err")


def init_from_env(backend=None):
    """(rank, local_rank, world, dist-or-None) from the torchrun environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, local_rank, world, dist


def _device(dist):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def max_over_ranks(x, dist):
    if dist is None:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, dist):
    if dist is None:
        return int(x)
    import torch
    t = torch.tensor([int(x)], dtype=torch.int64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


# ---- the packed u32 record of a clustered batch -----------------------------------------------------------------
def _words(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    return a.view(np.uint32).reshape(-1)


def pack_clustered(cb: ClusteredBatch, with_mindb=False, with_minimizers=True) -> np.ndarray:
    """Flat u32 image: header | per-representative arrays | [minimizer lists] | membership | [MinDB] | sequences."""
    rv = cb.rep_view
    n = cb.n_clusters
    of, orv = np.asarray(rv["off_fwd"], np.int64), np.asarray(rv["off_rev"], np.int64)
    nf, nr = (of[1:] - of[:-1]).astype(np.uint32), (orv[1:] - orv[:-1]).astype(np.uint32)
    ftot, rtot = int(of[-1] - of[0]), int(orv[-1] - orv[0])
    if with_minimizers:
        mv, mp = np.asarray(rv["min_val"], np.uint32), np.asarray(rv["min_pos"], np.uint32)
        assert int(of[0]) == 0 and int(orv[0]) == ftot and len(mv) == ftot + rtot, "records must be compact, forward lists first"
    else:
        mv = mp = np.zeros(0, np.uint32)
    keys, offs, post = cb.mindb if with_mindb else (np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.uint32))
    seq = np.frombuffer(cb.rep_seq, np.uint8) if cb.rep_seq is not None else np.zeros(0, np.uint8)
    seq_pad = np.zeros((len(seq) + 3) // 4 * 4, np.uint8)
    seq_pad[:len(seq)] = seq
    nm = len(cb.member_cls)
    head = np.array([_MAGIC, n, ftot, rtot, nm, len(keys), len(post), len(seq), 1 if cb.rep_seq is not None else 0,
                     cb.depth & 0xFFFFFFFF, 1 if with_minimizers else 0, 0], np.uint32)
    parts = [head, _words([cb.batch_start, cb.batch_end], np.int64), _words([rv.get("min_qual", 7.0)], np.float64), nf, nr]
    parts += [_words(rv[k], np.uint32) for k in _U32_FIELDS]
    parts += [_words(rv[k], np.float64) for k in _F64_FIELDS]
    parts += [mv, mp]
    parts += [_words(cb.member_cls, np.int32), _words(cb.member_read, np.int64), _words(cb.member_strand, np.int32)]
    parts += [_words(keys, np.uint32), _words(offs, np.int64), _words(post, np.uint32)]
    if cb.rep_seq is not None:
        parts += [_words(cb.rep_off, np.int64), seq_pad.view(np.uint32)]
    return np.concatenate(parts)


def unpack_clustered(buf: np.ndarray) -> ClusteredBatch:
    buf = np.ascontiguousarray(buf, np.uint32)
    magic, n, ftot, rtot, nm, nk, npost, nseq, has_seq, depth, has_min = (int(x) for x in buf[:11])
    if magic != _MAGIC:
        raise ValueError("not a packed clustered batch")
    pos = [12]

    def take(count, dtype):
        w = count * np.dtype(dtype).itemsize // 4
        a = buf[pos[0]:pos[0] + w].view(dtype).copy()
        pos[0] += w
        return a

    bs, be = (int(x) for x in take(2, np.int64))
    min_qual = float(take(1, np.float64)[0])
    nf, nr = take(n, np.uint32).astype(np.int64), take(n, np.uint32).astype(np.int64)
    rv = {k: take(n, np.uint32) for k in _U32_FIELDS}
    rv["state"] = rv["state"].astype(np.uint8)
    rv.update({k: take(n, np.float64) for k in _F64_FIELDS})
    nmin = ftot + rtot if has_min else 0
    rv["min_val"], rv["min_pos"] = take(nmin, np.uint32), take(nmin, np.uint32)
    off_f = np.zeros(n + 1, np.int64)
    off_f[1:] = np.cumsum(nf)
    off_r = np.zeros(n + 1, np.int64)
    off_r[1:] = np.cumsum(nr)
    rv["off_fwd"], rv["off_rev"], rv["min_qual"] = off_f, off_r + ftot, min_qual
    mc, mr, ms = take(nm, np.int32), take(nm, np.int64), take(nm, np.int32)
    keys, offs, post = take(nk, np.uint32), take(nk + 1, np.int64), take(npost, np.uint32)
    rep_seq = rep_off = None
    if has_seq:
        rep_off = take(n + 1, np.int64)
        rep_seq = take((nseq + 3) // 4, np.uint32).view(np.uint8)[:nseq].tobytes()
    return ClusteredBatch(rep_view=rv, member_cls=mc, member_read=mr, member_strand=ms, mindb=(keys, offs, post),
                          depth=depth if depth < 0x80000000 else depth - (1 << 32), batch_start=bs, batch_end=be,
                          rep_seq=rep_seq, rep_off=rep_off)


def _allgather_u32(buf_dev_or_np, dist, torch, dev):
    """Ragged all-gather of one u32 buffer per rank: returns (gathered tensor / array of world * cap words, cap, sizes).
    `buf_dev_or_np`: a torch int32 tensor on `dev` (used in place: RCCL moves HBM to HBM) or a numpy u32 array."""
    world = dist.get_world_size()
    is_t = torch.is_tensor(buf_dev_or_np)
    n = int(buf_dev_or_np.numel()) if is_t else len(buf_dev_or_np)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
    sz = [int(x) for x in sizes.cpu().numpy()]
    cap = max(max(sz), 1)
    if is_t:
        t = buf_dev_or_np
        if dev.type != t.device.type:      # (gloo in the CPU / one-GPU tests: the collective runs on host tensors)
            t = t.to(dev)
        send = t if n == cap else torch.cat([t, torch.zeros(cap - n, dtype=torch.int32, device=dev)])
    else:
        pad = np.zeros(cap, np.uint32)
        pad[:n] = buf_dev_or_np
        send = torch.from_numpy(pad.view(np.int32)).to(dev)
    recv = torch.empty(world * cap, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    return recv, cap, sz


def allgather_clustered(cb: ClusteredBatch, dist):
    """ONE ragged all-gather of every rank's packed record (minimizer lists included, from host arrays): the path
    without device-resident records.  Returns (batches, bytes per rank)."""
    if dist is None:
        return [cb], [0]
    import torch
    dev = _device(dist)
    recv, cap, sz = _allgather_u32(pack_clustered(cb), dist, torch, dev)
    out = recv.cpu().numpy().view(np.uint32)
    return [unpack_clustered(out[r * cap:r * cap + sz[r]]) for r in range(len(sz))], [4 * x for x in sz]


def _assemble(batches, cls, strand, st, rep_view=None, rep_seq=None, rep_off=None, mindb=None):
    """Membership of the merged clustering from the per-representative decisions of the one-pass merge
    (cluster.cpp:223-261: the members of a right cluster move with their strands flipped on a reverse-strand match)."""
    shift = np.cumsum([0] + [b.n_clusters for b in batches[:-1]])
    mcl = np.concatenate([cls[b.member_cls + s] for b, s in zip(batches, shift)]).astype(np.int32)
    mst = np.concatenate([strand[b.member_cls + s].astype(np.int32) * b.member_strand for b, s in zip(batches, shift)]).astype(np.int32)
    mrd = np.concatenate([b.member_read for b in batches])
    keep = mcl >= 0
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.uint32))
    if rep_view is None:   # only the counts of the merged clusters are known here
        rep_view = dict(hpc_len=np.zeros(int(st["n_clusters"]), np.uint32))
    return ClusteredBatch(rep_view=rep_view, rep_seq=rep_seq, rep_off=rep_off, member_cls=mcl[keep], member_read=mrd[keep],
                          member_strand=mst[keep], mindb=mindb if mindb is not None else empty, depth=batches[0].depth + len(batches) - 1,
                          batch_start=batches[0].batch_start, batch_end=batches[-1].batch_end, stats=st)


def _meta_view(batches):
    """Per-representative host arrays of all batches in merge order + the is_cluster mask of the leftmost batch."""
    v = {k: np.concatenate([np.asarray(b.rep_view[k]) for b in batches]) for k in VIEW_KEYS}
    v["min_qual"] = batches[0].rep_view.get("min_qual", 7.0)
    v["is_cluster"] = np.concatenate([np.full(b.n_clusters, 1 if i == 0 else 0, np.uint8) for i, b in enumerate(batches)])
    v["depth"] = 0
    if all(b.rep_seq is not None for b in batches):
        v["raw_seq"] = b"".join(b.rep_seq for b in batches)
        offs, base = [np.zeros(1, np.int64)], 0
        for b in batches:
            offs.append(np.asarray(b.rep_off, np.int64)[1:] + base)
            base += int(b.rep_off[-1])
        v["raw_off"] = np.concatenate(offs)
    return v


def merge_all(ctx, params, batches, min_cls_size=3, export_mindb=True):
    """((b0 + b1) + b2) ... as ONE pass of the merge path (see the module docstring), from host arrays."""
    if len(batches) == 1:
        return batches[0]
    if any(b.depth != 0 for b in batches):
        return fold_merge(ctx, params, batches, min_cls_size)     # merged batches: the MinClsSize gate is per merge
    rv = batches[0].rep_view
    for b in batches[1:]:
        rv = concat_records(rv, b.rep_view)
    v = _meta_view(batches)
    v.update(off_fwd=rv["off_fwd"], off_rev=rv["off_rev"], min_val=rv["min_val"], min_pos=rv["min_pos"], min_cls_size=min_cls_size)
    cls, strand, st = ctx.cluster_merge(params, None, v)
    # the merged clusters' records: the representatives that opened (or kept) a cluster, in cluster order
    n = len(cls)
    own = np.nonzero(cls >= 0)[0]
    first = own[np.unique(cls[own], return_index=True)[1]]
    assert np.array_equal(cls[first], np.arange(len(first)))
    rep_view = gather_records(rv, first)
    rep_seq, rep_off = gather_seqs(v.get("raw_seq"), v.get("raw_off"), first)
    mindb = ctx.index_export() if export_mindb else None
    return _assemble(batches, cls, strand, st, rep_view, rep_seq, rep_off, mindb)


def fold_merge(ctx, params, batches, min_cls_size=3):
    """The reference's left fold over clustered batches, one merge at a time (cross-check of the one-pass forms)."""
    left = batches[0]
    for b in batches[1:]:
        left = cluster_merge(ctx, params, left, b, min_cls_size=min_cls_size)
    return left


# ---- device-resident representative records -------------------------------------------------------------------------
def gather_local(ctx, cb: ClusteredBatch, torch, dev):
    """The minimizer lists of this rank's cluster representatives, gathered on the device out of the batch's query
    arrays (still in HBM after the clustering call) into two torch int32 tensors."""
    if cb.rep_entry is None or cb.ctx_serial != ctx.serial:
        raise RuntimeError("gather_local: the context no longer holds the queries of this batch")
    rv = cb.rep_view
    words = int((rv["off_fwd"][-1] - rv["off_fwd"][0]) + (rv["off_rev"][-1] - rv["off_rev"][0]))
    mins = torch.empty(max(words, 1), dtype=torch.int32, device=dev)
    poss = torch.empty(max(words, 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    w, _, _ = ctx.gather_records_device(cb.rep_entry, mins.data_ptr(), poss.data_ptr(), words)
    assert w == words
    return mins[:words], poss[:words]


def merge_gathered(ctx, params, metas, recv_min, recv_pos, cap, min_cls_size=3, export_mindb=False):
    """Step 3 of the module docstring: metas[r] = rank r's record without minimizers, recv_min / recv_pos = the
    all-gathered device buffers (rank r's block at [r * cap, ...): its forward lists, then its reverse lists).
    The blocks are re-laid on the device as [all forward lists][all reverse lists] (one HBM-to-HBM copy), which is
    the CSR form ioc_batch_view takes, and used in place."""
    import torch
    ft = [int(np.asarray(b.rep_view["off_fwd"])[-1]) for b in metas]                           # forward words per rank
    rt = [int(np.asarray(b.rep_view["off_rev"])[-1]) - f for b, f in zip(metas, ft)]           # reverse words per rank
    parts_f = [slice(r * cap, r * cap + ft[r]) for r in range(len(metas))]
    parts_r = [slice(r * cap + ft[r], r * cap + ft[r] + rt[r]) for r in range(len(metas))]
    cmin = torch.cat([recv_min[x] for x in parts_f] + [recv_min[x] for x in parts_r])
    cpos = torch.cat([recv_pos[x] for x in parts_f] + [recv_pos[x] for x in parts_r])
    fbase = np.cumsum([0] + ft)
    rbase = np.cumsum([0] + rt) + fbase[-1]
    off_f = np.concatenate([np.asarray(b.rep_view["off_fwd"], np.int64)[:-1] + fbase[r] for r, b in enumerate(metas)] + [fbase[-1:]])
    off_r = np.concatenate([np.asarray(b.rep_view["off_rev"], np.int64)[:-1] - ft[r] + rbase[r] for r, b in enumerate(metas)] + [rbase[-1:]])
    v = _meta_view(metas)
    v.update(off_fwd=off_f, off_rev=off_r, min_val=cmin.data_ptr(), min_pos=cpos.data_ptr(), total=int(cmin.numel()),
             minimizers_on_device=True, min_cls_size=min_cls_size, _keepalive=(cmin, cpos))
    torch.cuda.synchronize()        # torch's stream wrote the buffers, the context's stream reads them
    cls, strand, st = ctx.cluster_merge(params, None, v)
    mindb = ctx.index_export() if export_mindb else None
    return _assemble(metas, cls, strand, st, mindb=mindb)


def _ctx_device(ctx, torch):
    """The context's OWN card as a torch device, and torch's current device of the calling thread set to it: HIP's current device
    is per thread (a fresh worker thread starts on device 0), so nothing here may rely on torch.cuda.current_device()."""
    idx = int(getattr(ctx, "device", 0))
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def merge_all_device(ctx, params, cb: ClusteredBatch, dist, torch, min_cls_size=3, export_mindb=False, timing=None):
    """The merge of all ranks' freshly clustered batches with device-resident representative records (module
    docstring, steps 1-3).  Every rank returns the merged clustering (membership + counts; the merged clusters'
    records stay in HBM)."""
    import time
    dev = _ctx_device(ctx, torch)
    t0 = time.perf_counter()
    mins, poss = gather_local(ctx, cb, torch, dev)                       # step 1: HBM -> HBM
    meta = pack_clustered(cb, with_minimizers=False)
    t1 = time.perf_counter()
    if dist is None:
        recv_min, recv_pos, cap, metas, nbytes = mins, poss, int(mins.numel()), [unpack_clustered(meta)], [0]
    else:
        cdev = _device(dist)                                             # nccl: the GPU; gloo (tests): host
        recv_min, cap, sz = _allgather_u32(mins, dist, torch, cdev)      # step 2: RCCL all-gather of the device buffers
        recv_pos, _, _ = _allgather_u32(poss, dist, torch, cdev)
        rmeta, mcap, msz = _allgather_u32(meta, dist, torch, cdev)
        mh = rmeta.cpu().numpy().view(np.uint32)
        metas = [unpack_clustered(mh[r * mcap:r * mcap + msz[r]]) for r in range(len(msz))]
        nbytes = [8 * a + 4 * b for a, b in zip(sz, msz)]
        if cdev.type != "cuda":
            recv_min, recv_pos = recv_min.to(dev), recv_pos.to(dev)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    # step 3 with the work of the pass shared out over the ranks (ioc_set_shard, the exchange over this process group): fast mode
    # shards scoring and resolve by query, sahlin / furious shard their alignment rounds by query; IOC_DIST_SHARD=0: replicated
    import os
    world = dist.get_world_size() if dist is not None else 1
    shard = world > 1 and os.environ.get("IOC_DIST_SHARD", "1") != "0"
    if shard:
        ctx.set_shard(world, dist.get_rank(), torch_exchange(ctx, dist, torch))
    try:
        merged = merge_gathered(ctx, params, metas, recv_min, recv_pos, cap, min_cls_size, export_mindb)
        info = dict(sharded=shard, exchanges=ctx.shard_exchanges, aligned_pairs_this_rank=ctx.shard_aligned_pairs)
    finally:
        if shard:
            ctx.set_shard(1, 0, None)
    t3 = time.perf_counter()
    if timing is not None:
        timing.update(gather_ms=(t1 - t0) * 1e3, allgather_ms=(t2 - t1) * 1e3, merge_ms=(t3 - t2) * 1e3, payload_bytes_per_rank=nbytes,
                      clusters_in=[m.n_clusters for m in metas], **info)
    return merged


def timed_merge(ctx, params, cb: ClusteredBatch, dist, torch=None, dev=None):
    """bench.py's merge leg: device gather + all-gather + one-pass merge on every rank, each phase timed (max over
    ranks)."""
    from .digest import fnv1a_reads
    tm = {}
    merged = merge_all_device(ctx, params, cb, dist, torch, timing=tm)
    return {"batches": len(tm["clusters_in"]), "clusters_in": tm["clusters_in"], "clusters_out": merged.n_clusters,
            "reads_assigned": int(len(merged.member_read)), "gather_ms": max_over_ranks(tm["gather_ms"], dist),
            "allgather_ms": max_over_ranks(tm["allgather_ms"], dist), "merge_ms": max_over_ranks(tm["merge_ms"], dist),
            "payload_bytes_per_rank": tm["payload_bytes_per_rank"], "fnv1a": fnv1a_reads(merged),
            "merged_on": "every rank holds the gathered representatives; the pass's work is shared out by query (fast: scoring + "
                         "resolve, `valid` all-reduced per sweep; sahlin / furious: the alignment rounds, verdicts summed)" if tm.get("sharded")
                         else "every rank (device-resident representative records, replicated one-pass merge)",
            "replicated": not tm.get("sharded"), "exchanges_last_resolve": tm.get("exchanges"),
            "aligned_pairs_per_rank_max": max_over_ranks(tm.get("aligned_pairs_this_rank") or 0, dist),
            "aln_pairs_total": merged.stats.get("n_aln_pairs"),
            "aln_invoked": merged.stats.get("n_aln_invoked")}


def torch_exchange(ctx, dist, torch):
    """An ioc_set_shard exchange over a torch.distributed group, for transports other than the library's RCCL binding (and
    for two test ranks on ONE card, which RCCL refuses): the buffer is wrapped in place through __cuda_array_interface__;
    with gloo it is reduced on the host.  The context's stream is drained before and torch's after, so the order the C ABI
    asks for (after the work already on the stream, before what follows) holds."""
    from . import _lib
    dev = _ctx_device(ctx, torch)
    on_host = _device(dist).type != "cuda"

    class Wrap:
        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    def fn(ptr, count, kind, stream):
        if count <= 0:
            return 0
        ctx.synchronize()
        t = torch.as_tensor(Wrap(ptr, count, "|u1" if kind == _lib.XCHG_MAX_U8 else "<i4"), device=dev)
        if kind == _lib.XCHG_MAX_U8:
            w, op = t.clone(), dist.ReduceOp.MAX
        elif kind == _lib.XCHG_MIN_U32:
            w, op = t.to(torch.int64) & 0xFFFFFFFF, dist.ReduceOp.MIN           # unsigned order
        else:
            w, op = t.clone(), dist.ReduceOp.SUM
        if on_host:
            w = w.cpu()
        dist.all_reduce(w, op=op)
        t.copy_(w.to(dev).to(t.dtype) if kind != _lib.XCHG_MIN_U32 else (w.to(dev) & 0xFFFFFFFF).to(torch.int32))
        torch.cuda.synchronize()
        return 0
    return fn


# ---- the same exchange through the library's own C++ / RCCL binding (ioc_dist_*, csrc/ioc_dist.cpp) --------------------------
def native_unique_id(ctx, dist=None, torch=None):
    """The 128 bytes a communicator is made of: rank 0 draws them (ncclGetUniqueId in the library), the host program ships
    them — here through torch.distributed when the ranks were started by torchrun, nowhere at all for a single rank."""
    import ctypes as C
    from . import _lib
    rank = dist.get_rank() if dist is not None else 0
    ident = np.zeros(_lib_id_bytes(), np.uint8)
    if rank == 0:
        rc = ctx.L.ioc_dist_unique_id(ident.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise _lib.IocError(rc, "ioc_dist_unique_id failed")
    if dist is not None:
        dev = _device(dist)
        t = torch.from_numpy(ident).to(dev)
        dist.broadcast(t, src=0)
        ident = t.cpu().numpy().copy()
    return ident


def native_init(ctx, dist=None, torch=None, ident=None):
    """One RCCL communicator per context, made by the LIBRARY (ncclCommInitRank in C++).  ident: the id from
    native_unique_id, when the caller wants the torch collective that ships it on another thread than the init."""
    import ctypes as C
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if ident is None:
        ident = native_unique_id(ctx, dist, torch)
    ctx._chk(ctx.L.ioc_dist_init(ctx.h, ident.ctypes.data_as(C.POINTER(C.c_uint8)), rank, world))
    return rank, world


def _lib_id_bytes():
    return 128   # IOC_DIST_ID_BYTES


def merge_all_native(ctx, params, cb: ClusteredBatch, torch=None, min_cls_size=3, export_mindb=False, timing=None):
    """merge_all_device with every collective made by the library's C++ binding over RCCL (ioc_dist_merge): device gather of
    this rank's representatives' lists -> ragged all-gather HBM to HBM straight into the query layout -> ONE pass of the
    merge path on every rank.  Only the reads' membership (the host program's bookkeeping) is exchanged from here, through
    ioc_dist_allgatherv_host.  Needs native_init."""
    import ctypes as C
    import time
    from . import _lib
    from .api import _p
    L = ctx.L
    t0 = time.perf_counter()
    rank, world = C.c_int32(0), C.c_int32(1)
    ctx._chk(L.ioc_dist_info(ctx.h, C.byref(rank), C.byref(world)))
    W = world.value
    rv = dict(cb.rep_view)
    keep = None
    path = "host arrays (H2D inside ioc_dist_merge)"
    if torch is not None and cb.rep_entry is not None and cb.ctx_serial == ctx.serial:
        dev = _ctx_device(ctx, torch)
        mins, poss = gather_local(ctx, cb, torch, dev)                   # the lists never visit the host
        rv.update(min_val=mins.data_ptr(), min_pos=poss.data_ptr(), total=int(mins.numel()), minimizers_on_device=True)
        keep = (mins, poss)
        path = "device gather (lists HBM to HBM)"
    if cb.rep_seq is not None:
        rv.update(raw_seq=cb.rep_seq, raw_off=cb.rep_off)
    view, n, alive = ctx._make_view(rv)
    counts = np.zeros(W, np.int64)
    st = _lib.ClusterStats()
    tms = _lib.DistMergeTimes()
    ctx._chk(L.ioc_dist_merge(ctx.h, C.byref(params), _lib.TABLE_PATH.encode(), C.byref(view), min_cls_size, 0, None, None,
                              _p(counts, C.c_int64), None, None))
    N = int(counts.sum())
    cls, strand = np.zeros(max(N, 1), np.int32), np.zeros(max(N, 1), np.int8)
    t1 = time.perf_counter()
    ctx._chk(L.ioc_dist_merge(ctx.h, C.byref(params), _lib.TABLE_PATH.encode(), C.byref(view), min_cls_size, N,
                              _p(cls, C.c_int32), _p(strand, C.c_int8), _p(counts, C.c_int64), C.byref(st), C.byref(tms)))
    t2 = time.perf_counter()
    del keep, alive
    cls, strand = cls[:N], strand[:N]
    # membership of every rank's reads (cluster.cpp:223-261: members move with their representative)
    mine = np.concatenate([_words(cb.member_cls, np.int32), _words(cb.member_read, np.int64), _words(cb.member_strand, np.int32),
                           _words([cb.batch_start, cb.batch_end, cb.depth], np.int64)])
    sizes = np.zeros(W, np.int64)
    ctx._chk(L.ioc_dist_allgatherv_host(ctx.h, mine.ctypes.data_as(C.c_void_p), mine.nbytes, None, _p(sizes, C.c_int64)))
    allm = np.zeros(int(sizes.sum()) // 4, np.uint32)
    ctx._chk(L.ioc_dist_allgatherv_host(ctx.h, mine.ctypes.data_as(C.c_void_p), mine.nbytes, allm.ctypes.data_as(C.c_void_p),
                                        _p(sizes, C.c_int64)))
    metas, pos = [], 0
    for r in range(W):
        w = int(sizes[r]) // 4
        blk = allm[pos:pos + w]
        pos += w
        nm = (w - 6) // 4
        mc, mr = blk[:nm].view(np.int32), blk[nm:3 * nm].view(np.int64)
        ms = blk[3 * nm:4 * nm].view(np.int32)
        bs, be, dp = (int(x) for x in blk[4 * nm:].view(np.int64))
        metas.append(ClusteredBatch(rep_view=dict(hpc_len=np.zeros(int(counts[r]), np.uint32)), member_cls=mc, member_read=mr,
                                    member_strand=ms, mindb=None, depth=dp, batch_start=bs, batch_end=be))
    mindb = ctx.index_export() if export_mindb else None
    merged = _assemble(metas, cls, strand, st.as_dict(), mindb=mindb)
    if timing is not None:
        timing.update(sizing_ms=(t1 - t0) * 1e3, call_ms=(t2 - t1) * 1e3, exchange_lists_ms=float(tms.ms_exchange_lists),
                      merge_ms=float(tms.ms_merge), bytes_lists=int(tms.bytes_lists), bytes_records=int(tms.bytes_records),
                      clusters_in=[int(x) for x in counts], sharded=int(tms.sharded), exchanges=int(tms.exchanges), lists_path=path)
    return merged
