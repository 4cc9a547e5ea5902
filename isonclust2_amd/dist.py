"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Initial clustering shards batches one per GPU and needs no communication (the reference pipeline
runs them as separate processes, README.md:105-117).  The only exchange step of the path is the
merge: each rank contributes its clustered batch — representative records (minimizer SoA of every
cluster representative), membership and MinDB — through ONE ragged all-gather, after which the
reference's left fold ((b0 + b1) + b2) ... is replayed with ioc_cluster_merge.
"""
import io
import os

import numpy as np

from .pipeline import ClusteredBatch, cluster_merge

_REC = ("off_fwd", "off_rev", "min_val", "min_pos", "raw_len", "hpc_len", "score", "raw_err", "hpc_err", "state")


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


def pack_clustered(cb: ClusteredBatch) -> np.ndarray:
    """Flat byte image of a clustered batch (representative records + membership + MinDB)."""
    bio = io.BytesIO()
    arrs = {f"rep_{k}": np.asarray(cb.rep_view[k]) for k in _REC}
    arrs.update(member_cls=cb.member_cls, member_read=cb.member_read, member_strand=cb.member_strand,
                db_keys=cb.mindb[0], db_offs=cb.mindb[1], db_post=cb.mindb[2],
                meta=np.array([cb.depth, cb.batch_start, cb.batch_end], np.int64),
                min_qual=np.array([cb.rep_view.get("min_qual", 7.0)], np.float64))
    np.savez(bio, **arrs)
    return np.frombuffer(bio.getvalue(), np.uint8)


def unpack_clustered(buf: np.ndarray) -> ClusteredBatch:
    z = np.load(io.BytesIO(buf.tobytes()))
    rv = {k: z[f"rep_{k}"] for k in _REC}
    rv["min_qual"] = float(z["min_qual"][0])
    d, s, e = (int(x) for x in z["meta"])
    return ClusteredBatch(rep_view=rv, member_cls=z["member_cls"], member_read=z["member_read"],
                          member_strand=z["member_strand"], mindb=(z["db_keys"], z["db_offs"], z["db_post"]),
                          depth=d, batch_start=s, batch_end=e)


def allgather_clustered(cb: ClusteredBatch, dist):
    """Ragged all-gather of every rank's clustered batch (sizes first, then one padded payload)."""
    if dist is None:
        return [cb]
    import torch
    dev = _device(dist)
    world = dist.get_world_size()
    buf = pack_clustered(cb)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([len(buf)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine)
    cap = int(sizes.max().item())
    pad = np.zeros(cap, np.uint8)
    pad[:len(buf)] = buf
    send = torch.from_numpy(pad).to(dev)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy()
    return [unpack_clustered(out[r * cap:r * cap + int(sizes[r].item())]) for r in range(world)]


def fold_merge(ctx, params, batches, min_cls_size=3):
    """The reference's left fold over clustered batches, on one GPU."""
    left = batches[0]
    for b in batches[1:]:
        left = cluster_merge(ctx, params, left, b, min_cls_size=min_cls_size)
    return left
