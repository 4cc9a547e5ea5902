"""world_size = 2 over gloo on the CPU: the N > 1 plumbing of bench.py / dist.py (rank seeds, max / sum
reductions, the ragged all-gather of clustered batches).  The per-rank clustering itself needs a GPU;
here each rank's clustered batch comes from the oracle so that the exchanged payload is realistic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_clustered(seed):
    from isonclust2_amd import pipeline, synth
    from tests.helpers import oracle_sorted_batch
    rs = synth.generate_config("tiny", seed=seed)
    B, view = oracle_sorted_batch(rs)
    B.cluster(mode="fast")
    acl, ast = B.assignments(rs.n)
    ent_cls, ent_strand = acl[view["orig"]], ast[view["orig"]]
    ok = ent_cls >= 0
    reps = []
    seen = set()
    for i in np.nonzero(ok)[0]:
        if int(ent_cls[i]) not in seen:
            seen.add(int(ent_cls[i]))
            reps.append(i)
    keys, offs, post = B.index()
    return pipeline.ClusteredBatch(rep_view=pipeline.gather_records(view, np.array(reps)),
                                   member_cls=ent_cls[ok].astype(np.int32),
                                   member_read=view["orig"][ok].astype(np.int64),
                                   member_strand=ent_strand[ok].astype(np.int32), mindb=(keys, offs, post),
                                   depth=0, batch_start=0, batch_end=rs.n - 1), rs.n


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from isonclust2_amd import dist as d
    r, lr, w, dist = d.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist is not None
    cb, n = _oracle_clustered(seed=1 + rank)      # bench.py: seed = 1 + rank, one batch per rank
    elapsed = 0.5 + 0.25 * rank
    tmax = d.max_over_ranks(elapsed, dist)
    total = d.sum_over_ranks(n, dist)
    got, nbytes = d.allgather_clustered(cb, dist)
    assert len(nbytes) == world and min(nbytes) > 0
    ok = len(got) == world
    for rr, g in enumerate(got):
        exp, _ = _oracle_clustered(seed=1 + rr)
        ok &= np.array_equal(g.member_cls, exp.member_cls) and np.array_equal(g.member_read, exp.member_read)
        ok &= len(g.mindb[2]) == 0     # no MinDB travels: the merge rebuilds the left index from b0's representatives
        ok &= np.array_equal(g.rep_view["min_val"], exp.rep_view["min_val"]) and np.array_equal(g.rep_view["hpc_err"], exp.rep_view["hpc_err"])
        ok &= np.array_equal(g.rep_view["off_rev"], exp.rep_view["off_rev"]) and g.batch_end == exp.batch_end
    q.put((rank, tmax, total, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tmax, total, ok in res:
        assert tmax == pytest.approx(0.75)   # MAX over ranks
        assert total == 128                  # 64 reads per rank
        assert ok


def test_membership_of_a_merge_from_per_representative_decisions():
    """dist._assemble (the host bookkeeping of the one-pass merge, cluster.cpp:223-261) against the oracle folding two
    batches of ONE read set (joins across the batches, strand flips): the per-representative decisions are read off
    the oracle's result, the membership of every read must come out as the oracle's."""
    from isonclust2_amd import dist as d
    from isonclust2_amd import pipeline, synth
    from oracle import pyoracle as po
    rs = synth.generate_config("config1", seed=5)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    p = po.default_params(11, 15)
    h = rs.n // 2
    obs = [po.Batch(R, 0, h - 1, p, 0), po.Batch(R, h, rs.n - 1, p, 1)]
    cbs = []
    for B in obs:
        B.cluster(mode="fast")
        cls, orig, strand, is_rep = B.members()
        real = is_rep == 0
        ncl = B.n_clusters()
        cbs.append(pipeline.ClusteredBatch(rep_view=dict(hpc_len=np.zeros(ncl, np.uint32)), member_cls=cls[real].astype(np.int32),
                                           member_read=orig[real].astype(np.int64), member_strand=strand[real].astype(np.int32),
                                           mindb=(None, None, None), depth=0, batch_start=0, batch_end=rs.n - 1))
    first_read = [np.array([b.member_read[np.nonzero(b.member_cls == c)[0][0]] for c in range(b.n_clusters)]) for b in cbs]
    first_strand = [np.array([b.member_strand[np.nonzero(b.member_cls == c)[0][0]] for c in range(b.n_clusters)]) for b in cbs]
    obs[0].cluster(right=obs[1], mode="fast")
    ocl, ost = obs[0].assignments(rs.n)
    # decisions per representative, in merge order (b0's clusters, then b1's): the merged id and the strand factor
    cls = np.concatenate([ocl[first_read[0]], ocl[first_read[1]]]).astype(np.int32)
    strand = np.concatenate([ost[first_read[0]] * first_strand[0], ost[first_read[1]] * first_strand[1]]).astype(np.int8)
    assert (strand[:cbs[0].n_clusters] == 1).all() and (strand[cbs[0].n_clusters:] == -1).any()
    merged = d._assemble(cbs, cls, strand, {"n_clusters": obs[0].n_clusters()})
    mcl, mst = merged.assignments(rs.n)
    assert np.array_equal(mcl, ocl) and np.array_equal(mst, ost)
    assert merged.n_clusters == obs[0].n_clusters()


def test_pack_roundtrip_and_record_concat():
    from isonclust2_amd import dist as d
    from isonclust2_amd import pipeline
    cb, _ = _oracle_clustered(seed=3)
    g = d.unpack_clustered(d.pack_clustered(cb))
    for k in ("off_fwd", "off_rev", "min_val", "min_pos", "hpc_len", "hpc_err"):
        assert np.array_equal(g.rep_view[k], cb.rep_view[k])
    m = d.unpack_clustered(d.pack_clustered(cb, with_minimizers=False))      # what travels beside the device buffers
    assert len(m.rep_view["min_val"]) == 0 and np.array_equal(m.rep_view["off_rev"], cb.rep_view["off_rev"])
    assert np.array_equal(m.member_read, cb.member_read) and len(d.pack_clustered(cb, with_minimizers=False)) < len(d.pack_clustered(cb)) // 4
    # gather_records / concat_records keep every representative's lists intact
    n = cb.n_clusters
    a = pipeline.gather_records(cb.rep_view, np.arange(0, n // 2))
    b = pipeline.gather_records(cb.rep_view, np.arange(n // 2, n))
    c = pipeline.concat_records(a, b)
    for i in range(n):
        for off in ("off_fwd", "off_rev"):
            s0, e0 = cb.rep_view[off][i], cb.rep_view[off][i + 1]
            s1, e1 = c[off][i], c[off][i + 1]
            assert np.array_equal(cb.rep_view["min_val"][s0:e0], c["min_val"][s1:e1])
            assert np.array_equal(cb.rep_view["min_pos"][s0:e0], c["min_pos"][s1:e1])
