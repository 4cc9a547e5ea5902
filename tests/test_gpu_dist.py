"""Two ranks on the GPU box (both on GPU 0, gloo for the collective — RCCL refuses two ranks on one device): the N > 1
path of bench.py end to end — every rank clusters its own batch through the C ABI, gathers its representative records
on the device, all-gathers them, merges in one pass — against the oracle folding the same batches."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.zeros(1, device="cuda:0")
    from isonclust2_amd import api, dist as d, pipeline, synth
    from isonclust2_amd.digest import fnv1a_reads
    _, _, _, dist = d.init_from_env(backend="gloo")
    rs = synth.generate_config("config1", seed=6)
    ctx = api.Context(0)
    # the read set sorted once (as `sort` does), cut into `world` consecutive batches: rank r clusters batch r
    sb_all, order = pipeline.sort_stage(ctx, rs, 11, 15)
    cuts = np.linspace(0, rs.n, world + 1).astype(int)
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    v = sb_all.view
    sel = np.arange(a, b)
    view = pipeline.gather_records(v, sel)
    seq, off = pipeline.gather_seqs(v["raw_seq"], v["raw_off"], sel)
    view.update(raw_seq=seq, raw_off=off)
    sb = pipeline.SortedBatch(view=view, read_ids=sb_all.read_ids[a:b], batch_nr=rank, batch_start=a, batch_end=b - 1)
    p = api.default_params(11, 15, mode)
    cb = pipeline.cluster_single(ctx, p, sb)
    res = d.timed_merge(ctx, p, cb, dist, torch, torch.device("cuda", 0))
    q.put((rank, res["fnv1a"], res["clusters_out"], res["reads_assigned"], res["payload_bytes_per_rank"]))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("mode", ["fast", "sahlin"])
def test_two_ranks_cluster_and_merge(mode):
    from isonclust2_amd import synth
    from isonclust2_amd.digest import fnv1a
    from oracle import pyoracle as po
    world = 2
    rs = synth.generate_config("config1", seed=6)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    cuts = np.linspace(0, rs.n, world + 1).astype(int)
    obs = [po.Batch(R, int(cuts[r]), int(cuts[r + 1]) - 1, po.default_params(11, 15), r) for r in range(world)]
    for B in obs:
        B.cluster(mode=mode)
    obs[0].cluster(right=obs[1], mode=mode)
    ocl, ost = obs[0].assignments(rs.n)
    want = f"{fnv1a(ocl, ost):016x}"
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    ps = [mpc.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, dig, ncl, nreads, nbytes in res:
        assert dig == want, (rank, dig, want)                 # every rank holds the same merged clustering = the oracle's fold
        assert ncl == obs[0].n_clusters() and nreads == int((ocl >= 0).sum())
        assert len(nbytes) == world


@pytest.mark.parametrize("mode", ["fast", "sahlin"])
def test_native_rccl_binding_single_rank(mode):
    """The library's own C++ / RCCL binding (csrc/ioc_dist.cpp) with the one rank a one-GPU box allows: ncclCommInitRank, the
    size exchange, the grouped-broadcast ragged all-gather HBM to HBM and the host-record exchange all execute; with one rank
    every representative is a cluster from the start, so the merged clustering must be the batch's own (against the oracle).
    The multi-rank semantics of the same one-pass merge are covered by merge_gathered (tests above, test_gpu_fullsize.py)."""
    import torch
    from isonclust2_amd import api, dist as d, pipeline, synth
    from isonclust2_amd.digest import fnv1a, fnv1a_reads
    from tests.helpers import oracle_entry_assignments, oracle_sorted_batch
    torch.zeros(1, device="cuda:0")
    rs = synth.generate_config("config1", seed=6)
    ctx = api.Context(0)
    sb, order = pipeline.sort_stage(ctx, rs, 11, 15)
    p = api.default_params(11, 15, mode)
    cb = pipeline.cluster_single(ctx, p, sb)
    assert d.native_init(ctx, None, torch) == (0, 1)
    tm = {}
    merged = d.merge_all_native(ctx, p, cb, torch, export_mindb=True, timing=tm)     # device-resident lists (gather_local)
    assert tm["clusters_in"] == [cb.n_clusters] and tm["bytes_lists"] > 0
    assert merged.n_clusters == cb.n_clusters and fnv1a_reads(merged, rs.n) == fnv1a_reads(cb, rs.n)
    for a, b in zip(merged.mindb, cb.mindb):
        assert np.array_equal(a, b)                                                  # the MinDB AddMinimizers makes of the same clusters
    merged2 = d.merge_all_native(ctx, p, cb, None)                                   # the same from host arrays (H2D inside)
    assert fnv1a_reads(merged2, rs.n) == fnv1a_reads(cb, rs.n)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, _ = oracle_entry_assignments(B, view, mode=mode)
    acl, ast = merged.assignments(rs.n)
    assert f"{fnv1a(acl[sb.read_ids], ast[sb.read_ids]):016x}" == f"{fnv1a(ocl, ost):016x}"
    # the collectives on their own
    import ctypes as C
    x = C.c_double(3.5)
    ctx._chk(ctx.L.ioc_dist_allreduce_max(ctx.h, C.byref(x)))
    assert x.value == 3.5
    ctx._chk(ctx.L.ioc_dist_barrier(ctx.h))
    ctx._chk(ctx.L.ioc_dist_shutdown(ctx.h))
    ctx.close()
