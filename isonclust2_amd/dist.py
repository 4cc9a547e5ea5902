"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Initial clustering shards batches one per GPU and needs no communication (the reference pipeline
runs them as separate processes, README.md:105-117).  The one exchange step of the path is the
merge (`cluster -l L -r R`, src/cluster.cpp:67-322 with two batches):

  1. every rank packs its clustered batch — one record per cluster representative (forward + reverse
     minimizer lists, lengths, error rates, raw sequence for sahlin / furious), the membership of its reads
     and, on rank 0 only, the MinDB of the leftmost batch — into ONE flat u32 buffer (`pack_clustered`);
  2. ONE ragged all-gather of those buffers as device tensors (RCCL over xGMI; sizes first);
  3. every rank replays the reference's left fold ((b0 + b1) + b2) ... on its own GPU (`merge_all`).

Step 3 is ONE pass of the merge path, not world-1 of them: with consensus off a merge only appends the right
batch's unmatched clusters to the left and extends the MinDB by their minimizers (cluster.cpp:178-217), so
folding b1, b2, ... one after the other makes exactly the decisions of one greedy loop over the concatenated
right representatives of b1, b2, ... against the left state of b0 (same order of queries, same left state at every
query, `right.Depth == 0` for freshly clustered batches so the MinClsSize gate of cluster.cpp:119-123 is off) —
checked against the oracle's step-by-step fold in tests/test_gpu_fullsize.py and tests/test_dist_gloo.py.
All-pairs scoring + fixed-point resolve of that one pass run on every rank (replicated: the result is needed
everywhere, and a rank's share of the candidate tables would be ~N^2 x 12 B to exchange — two orders of magnitude
more than the representative records themselves).
"""
import os
import time

import numpy as np

from .pipeline import ClusteredBatch, cluster_merge, concat_records

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


def pack_clustered(cb: ClusteredBatch, with_mindb=True) -> np.ndarray:
    """Flat u32 image: header | per-representative arrays | minimizer lists | membership | MinDB | sequences."""
    rv = cb.rep_view
    n = cb.n_clusters
    of, orv = np.asarray(rv["off_fwd"], np.int64), np.asarray(rv["off_rev"], np.int64)
    nf, nr = (of[1:] - of[:-1]).astype(np.uint32), (orv[1:] - orv[:-1]).astype(np.uint32)
    ftot, rtot = int(of[-1] - of[0]), int(orv[-1] - orv[0])
    mv, mp = np.asarray(rv["min_val"], np.uint32), np.asarray(rv["min_pos"], np.uint32)
    assert int(of[0]) == 0 and int(orv[0]) == ftot and len(mv) == ftot + rtot, "records must be compact, forward lists first"
    keys, offs, post = cb.mindb if with_mindb else (np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.uint32))
    seq = np.frombuffer(cb.rep_seq, np.uint8) if cb.rep_seq is not None else np.zeros(0, np.uint8)
    seq_pad = np.zeros((len(seq) + 3) // 4 * 4, np.uint8)
    seq_pad[:len(seq)] = seq
    nm = len(cb.member_cls)
    head = np.array([_MAGIC, n, ftot, rtot, nm, len(keys), len(post), len(seq), 1 if cb.rep_seq is not None else 0,
                     cb.depth & 0xFFFFFFFF, 0, 0], np.uint32)
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
    magic, n, ftot, rtot, nm, nk, npost, nseq, has_seq, depth = (int(x) for x in buf[:10])
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
    rv["min_val"], rv["min_pos"] = take(ftot + rtot, np.uint32), take(ftot + rtot, np.uint32)
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


def allgather_clustered(cb: ClusteredBatch, dist):
    """ONE ragged all-gather of every rank's packed record as u32 tensors on the backend's device (RCCL: HBM to HBM
    over xGMI; sizes first, then one padded payload).  Only rank 0's record carries a MinDB (the leftmost batch's)."""
    if dist is None:
        return [cb], [0]
    import torch
    dev = _device(dist)
    world, rank = dist.get_world_size(), dist.get_rank()
    buf = pack_clustered(cb, with_mindb=(rank == 0))
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([len(buf)], dtype=torch.int64, device=dev))
    cap = int(sizes.max().item())
    send = torch.zeros(cap, dtype=torch.int32, device=dev)
    send[:len(buf)] = torch.from_numpy(buf.view(np.int32)).to(dev)
    recv = torch.empty(world * cap, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy().view(np.uint32)
    sz = sizes.cpu().numpy()
    return [unpack_clustered(out[r * cap:r * cap + int(sz[r])]) for r in range(world)], [int(4 * x) for x in sz]


def concat_right(batches):
    """The right batches b1, b2, ... as ONE right batch: representative records, membership (cluster ids shifted),
    sequences."""
    rv = batches[0].rep_view
    for b in batches[1:]:
        rv = concat_records(rv, b.rep_view)
    shift = np.cumsum([0] + [b.n_clusters for b in batches[:-1]])
    have_seq = all(b.rep_seq is not None for b in batches)
    rep_seq = rep_off = None
    if have_seq:
        rep_seq = b"".join(b.rep_seq for b in batches)
        offs, base = [np.zeros(1, np.int64)], 0
        for b in batches:
            offs.append(np.asarray(b.rep_off, np.int64)[1:] + base)
            base += int(b.rep_off[-1])
        rep_off = np.concatenate(offs)
    return ClusteredBatch(rep_view=rv, rep_seq=rep_seq, rep_off=rep_off,
                          member_cls=np.concatenate([b.member_cls + s for b, s in zip(batches, shift)]).astype(np.int32),
                          member_read=np.concatenate([b.member_read for b in batches]),
                          member_strand=np.concatenate([b.member_strand for b in batches]),
                          mindb=(np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.uint32)), depth=0,
                          batch_start=batches[0].batch_start, batch_end=batches[-1].batch_end)


def merge_all(ctx, params, batches, min_cls_size=3):
    """((b0 + b1) + b2) ... as ONE pass of the merge path (see the module docstring)."""
    if len(batches) == 1:
        return batches[0]
    if any(b.depth != 0 for b in batches[1:]):
        return fold_merge(ctx, params, batches, min_cls_size)     # merged right batches: the MinClsSize gate is per merge
    return cluster_merge(ctx, params, batches[0], concat_right(batches[1:]), min_cls_size=min_cls_size)


def fold_merge(ctx, params, batches, min_cls_size=3):
    """The reference's left fold over clustered batches, one merge at a time."""
    left = batches[0]
    for b in batches[1:]:
        left = cluster_merge(ctx, params, left, b, min_cls_size=min_cls_size)
    return left


def timed_merge(ctx, params, cb: ClusteredBatch, dist, torch=None, dev=None):
    """bench.py's merge leg: pack + all-gather + one-pass merge on every rank, each phase timed (max over ranks)."""
    t0 = time.perf_counter()
    allb, nbytes = allgather_clustered(cb, dist)
    if torch is not None and dev is not None and dev.type == "cuda":
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    merged = merge_all(ctx, params, allb)
    t2 = time.perf_counter()
    n_reads = int(sum(len(b.member_read) for b in allb))
    from .digest import fnv1a_reads
    return {"batches": len(allb), "clusters_in": [b.n_clusters for b in allb], "clusters_out": merged.n_clusters,
            "reads_assigned": n_reads, "allgather_ms": max_over_ranks((t1 - t0) * 1e3, dist),
            "merge_ms": max_over_ranks((t2 - t1) * 1e3, dist),
            "payload_bytes_per_rank": nbytes,
            "fnv1a": fnv1a_reads(merged), "merged_on": "every rank (replicated one-pass merge)",
            "aln_invoked": merged.stats.get("n_aln_invoked")}
