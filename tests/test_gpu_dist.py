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
    # the three all-reduces of the sharded resolve (ioc_dist_set_shard) run on RCCL's own data types and operators
    from isonclust2_amd import _lib
    for kind, t in ((_lib.XCHG_MAX_U8, torch.arange(0, 200, dtype=torch.uint8, device="cuda:0")),
                    (_lib.XCHG_MIN_U32, torch.tensor([-1, 7, 0], dtype=torch.int32, device="cuda:0")),
                    (_lib.XCHG_SUM_I32, torch.tensor([-5, 9, 1 << 30], dtype=torch.int32, device="cuda:0"))):
        want = t.clone()
        torch.cuda.synchronize()
        ctx._chk(ctx.L.ioc_dist_exchange(ctx.h, t.data_ptr(), t.numel(), kind))
        ctx.synchronize()
        assert torch.equal(t, want)
    ctx._chk(ctx.L.ioc_dist_set_shard(ctx.h, 1))          # one rank: nothing to shard, the setting stays off
    cb1 = pipeline.cluster_single(ctx, p, sb)
    assert ctx.shard_exchanges == 0 and fnv1a_reads(cb1, rs.n) == fnv1a_reads(cb, rs.n)
    ctx._chk(ctx.L.ioc_dist_shutdown(ctx.h))
    ctx.close()


def _worker_shard(rank, world, port, q):
    """Every rank holds the SAME queries; score + resolve sharded by query (ioc_set_shard) against the unsharded run."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    import faulthandler
    import traceback
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", f"shard_rank{rank}.log"), "w", buffering=1)
    faulthandler.enable(log)
    try:
        import torch
        import torch.distributed as dist
        torch.zeros(1, device="cuda:0")
        from isonclust2_amd import api, dist as d, pipeline, synth
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=90))    # a lost peer fails the run, not hangs it
        ctx = api.Context(0)
        out = []

        def same(a, b):
            return (np.array_equal(a.member_cls, b.member_cls) and np.array_equal(a.member_strand, b.member_strand)
                    and np.array_equal(a.member_read, b.member_read) and all(np.array_equal(x, y) for x, y in zip(a.mindb, b.mindb)))

        def run(p, sb):
            half = len(sb.read_ids) // 2
            whole = pipeline.cluster_single(ctx, p, sb)
            it_whole = ctx.timings()["resolve_iters"]
            x0 = ctx.shard_exchanges          # all-reduces of the last resolve: 2 per sweep + 4 for the decisions and cuts
            print("  whole", whole.n_clusters, it_whole, x0, file=log)
            a = pipeline.cluster_single(ctx, p, pipeline.slice_sorted(sb, 0, half, batch_nr=0))
            b = pipeline.cluster_single(ctx, p, pipeline.slice_sorted(sb, half, len(sb.read_ids), batch_nr=1))
            merged = pipeline.cluster_merge(ctx, p, a, b)          # L > 0: queries against a left MinDB
            print("  merged", merged.n_clusters, file=log)
            return whole, merged, it_whole, x0

        for name, seed, rng in (("tiny", 1, None), ("short_dup", 2, None), ("config2", 3, None), ("short_dup", 5, "512")):
            # (IOC_SCORE_RANGE below the batch size: the scoring kernel of large merges, several target ranges per query)
            os.environ.pop("IOC_SCORE_RANGE", None)
            if rng:
                os.environ["IOC_SCORE_RANGE"] = rng
            rs = synth.generate_config(name, seed=seed)
            sb, _ = pipeline.sort_stage(ctx, rs, 11, 15)
            p = api.default_params(11, 15, "fast")
            ctx.set_shard(1, 0, None)
            print(name, "unsharded", file=log)
            w0, m0, it0, x0 = run(p, sb)
            ctx.set_shard(world, rank, d.torch_exchange(ctx, dist, torch))
            print(name, "sharded", file=log)
            w1, m1, it1, x1 = run(p, sb)
            ok = same(w0, w1) and same(m0, m1) and x0 == 0 and x1 >= 2 * it1 + 4 and it1 == it0
            out.append((name, ok, w0.n_clusters, w1.n_clusters, m0.n_clusters, m1.n_clusters, it0, it1, x0, x1))
            if name in ("tiny", "short_dup") and not rng:
                # sahlin / furious: scoring and resolve stay replicated, the ALIGNMENT rounds are shared out by owner of the query
                # and their verdicts summed over the ranks: same clustering, and this rank aligned only its share of the pairs
                for mode in ("sahlin", "furious"):
                    ps = api.default_params(11, 15, mode)
                    ctx.set_shard(1, 0, None)
                    s0 = pipeline.cluster_single(ctx, ps, sb)
                    pairs0 = ctx.timings()["n_align_pairs"]
                    half = len(sb.read_ids) // 2
                    a0 = pipeline.cluster_single(ctx, ps, pipeline.slice_sorted(sb, 0, half, batch_nr=0))
                    b0 = pipeline.cluster_single(ctx, ps, pipeline.slice_sorted(sb, half, len(sb.read_ids), batch_nr=1))
                    m0s = pipeline.cluster_merge(ctx, ps, a0, b0)
                    ctx.set_shard(world, rank, d.torch_exchange(ctx, dist, torch))
                    s1 = pipeline.cluster_single(ctx, ps, sb)
                    mine = ctx.shard_aligned_pairs
                    m1s = pipeline.cluster_merge(ctx, ps, a0, b0)       # (the merge's alignment rounds: queries against left representatives)
                    tot = torch.tensor([mine], dtype=torch.int64)
                    dist.all_reduce(tot)
                    st_pairs = s0.stats.get("n_aln_pairs", -1)
                    ok_s = same(s0, s1) and same(m0s, m1s) and int(tot.item()) == st_pairs and (st_pairs < 4 or 0 < mine < st_pairs)
                    print(f"  {mode}: pairs {st_pairs} (device count {pairs0}), this rank {mine}, all ranks {int(tot.item())}", file=log)
                    out.append((f"{name}/{mode}", ok_s, s0.n_clusters, s1.n_clusters, m0s.n_clusters, m1s.n_clusters, st_pairs, mine, int(tot.item()), 0))
                ctx.set_shard(1, 0, None)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
    except BaseException:
        traceback.print_exc(file=log)
        q.put((rank, [("worker failed", False, traceback.format_exc())]))
        raise


def test_sharded_score_and_resolve_two_ranks():
    """ioc_set_shard: two ranks on the one card, each scoring and deciding every second query and
    all-reducing `valid` after every sweep (gloo through dist.torch_exchange; ioc_dist_merge installs the RCCL form of the
    same hook), give the assignments and the MinDB of the unsharded run, in the same number of sweeps."""
    world = 2
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    ps = [mpc.Process(target=_worker_shard, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in ps:
        p.join(timeout=120)
    for rank, out in res:
        for rec in out:
            assert rec[1], (rank, rec)
    assert all(p.exitcode == 0 for p in ps)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sahlin", "fast"])
def test_scale_command_path(mode):
    """The command the driver's scaling run issues — `python bench.py --gpus N` — end to end with two ranks sharing this card over
    gloo (RCCL refuses two ranks on one device): the launcher that touches no GPU, torch.distributed.run, the ranks' batches,
    the timed region with its barrier, the merge of the ranks' representatives with its work shared out, rank 0's one JSON line.
    (the path's first execution must not be the first 8-GPU lease.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-cli", "--no-core", "--strict", "--mode", mode], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["reads_per_gpu"] == 3000 and j["config"]["identical_batches"] is False
    assert j["golden_parity"] and all(v.startswith("every rank") for v in j["golden_parity"].values()), j["golden_parity"]
    m = j["merge"]
    assert "error" not in m, m
    assert m["mode"] == mode and m["replicated"] is False, m
    assert m["golden"]["clusters_match"] is True, m["golden"]
    assert abs(j["value"] - 2 * 3000 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
