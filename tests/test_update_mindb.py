"""UpdateMinDB (src/minimizer.cpp:124-160).

CPU: the oracle's restatement against the semantics spelled out in the reference (std::set differences,
lists through a std::set on delete, push_back + sort on insert, `db[m]` creating entries, emptied lists kept).
GPU: ioc_index_update rewrites the device-resident MinDB (CSR) and the per-cluster value sets; the result
and a merge run on top of it equal the oracle's."""
import numpy as np
import pytest

from isonclust2_amd import synth
from oracle import pyoracle as po


def _py_update(db, best, old, new):
    """Plain-Python reading of minimizer.cpp:124-160 on a dict value -> list."""
    olds, news = set(int(x) for x in old), set(int(x) for x in new)
    for m in sorted(olds - news):
        lst = db.setdefault(m, [])
        db[m] = sorted(set(lst) - {best})
    for m in sorted(news - olds):
        lst = db.setdefault(m, [])
        lst.append(best)
        lst.sort()
    return db


def _index_dict(B):
    keys, offs, post = B.index()
    return {int(k): [int(x) for x in post[offs[i]:offs[i + 1]]] for i, k in enumerate(keys)}


def _clustered_batch(cfg="tiny", seed=3, k=11, w=15):
    rs = synth.generate_config(cfg, seed=seed)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    B = po.Batch(R, 0, rs.n - 1, po.default_params(k, w))
    B.cluster(mode="fast")
    return rs, B


def _rep_values(B, c):
    """forward minimizer values of cluster c's representative (entry c after clustering)."""
    mn, _, _ = B.entry_mins(c, 0, 100000)
    return np.asarray(mn, np.uint32)


def test_oracle_update_matches_the_reference_semantics():
    rs, B = _clustered_batch()
    ncl = B.n_clusters()
    assert ncl >= 3
    db = _index_dict(B)
    rng = np.random.default_rng(5)
    for step, c in enumerate([0, ncl - 1, 1, 0]):
        old = _rep_values(B, c)
        if step == 0:      # drop a third of the values, add unseen ones (new keys) and values of another cluster
            new = np.concatenate([old[::3], np.array([0xFFFFFFFF, 7, 7, 123456789], np.uint32), _rep_values(B, 2)[:50]])
        elif step == 1:    # nothing in common: every old list loses c
            new = rng.integers(0, 4 ** 11, 300, dtype=np.uint32)
        elif step == 2:    # identical: no-op
            new = old.copy()
        else:              # back to a permutation of values with duplicates
            new = np.concatenate([old, old[:10]])
        B.update_mindb(c, old, new)
        db = _py_update(db, c, old, new)
        assert _index_dict(B) == db, step
    assert any(len(v) == 0 for v in db.values()), "emptied lists stay in the index as keys"


def test_oracle_update_creates_entries_for_absent_old_values():
    """`auto& mins = db[m]` (minimizer.cpp:146): a toDel value that is not a key becomes an empty entry."""
    rs, B = _clustered_batch()
    db = _index_dict(B)
    ghost = np.array([0xFFFFFFF0, 0xFFFFFFF1], np.uint32)
    assert all(int(g) not in db for g in ghost)
    old = np.concatenate([_rep_values(B, 0), ghost])
    B.update_mindb(0, old, _rep_values(B, 0), mins_too=False)
    now = _index_dict(B)
    assert now[int(ghost[0])] == [] and now[int(ghost[1])] == []
    for kk, v in db.items():
        assert now[kk] == v


# ---- GPU ---------------------------------------------------------------------------------------------------
def _left_state(ctx, api, B):
    keys, offs, post = B.index()
    info, *_ = B.minimizer_soa()
    ncl = B.n_clusters()
    cells = np.array([api.host_err_cell(e) for e in info["hpc_err"][:ncl]], np.uint8)
    ctx.left_load(ncl, cells, keys, offs, post)
    return info


@pytest.mark.gpu
def test_device_update_equals_oracle():
    from isonclust2_amd import api
    ctx = api.Context(0)
    rs, B = _clustered_batch("config1", seed=2)
    _left_state(ctx, api, B)
    ncl = B.n_clusters()
    rng = np.random.default_rng(9)
    plan = [0, ncl // 2, ncl - 1, 0, 3]
    for step, c in enumerate(plan):
        old = _rep_values(B, c)
        other = _rep_values(B, (c + 5) % ncl)
        if step % 3 == 0:
            new = np.concatenate([old[::2], other[: len(other) // 2], np.array([0xFFFFFFFF, 5], np.uint32)])
        elif step % 3 == 1:
            new = rng.integers(0, 4 ** 11, 500, dtype=np.uint32)
        else:
            new = np.concatenate([old, other])
        B.update_mindb(c, old, new)
        ctx.index_update(c, old, new)
        keys, offs, post = B.index()
        dk, do, dp = ctx.left_export()
        assert np.array_equal(dk, keys), step
        assert np.array_equal(do, offs), step
        assert np.array_equal(dp, post), step
    # old minimizers that are not what the index holds are refused (the reference never passes such)
    with pytest.raises(Exception):
        ctx.index_update(1, _rep_values(B, 2), _rep_values(B, 1))
    # no-op update
    v = _rep_values(B, 1)
    ctx.index_update(1, v, v)
    assert np.array_equal(ctx.left_export()[2], B.index()[2])
    ctx.close()


@pytest.mark.gpu
def test_merge_on_top_of_a_device_update_equals_oracle():
    """Left batch clustered, three of its representatives replaced (values of other reads: what a consensus
    would do), then the right batch merged in — on the device against the updated resident left state."""
    from isonclust2_amd import api, pipeline
    from tests.test_gpu_merge import _batches
    ctx = api.Context(0)
    rs = synth.generate_config("config1", seed=4)
    obs, sbs = _batches(rs, 2)
    p = api.default_params(11, 15, "fast")
    for Bo in obs:
        Bo.cluster(mode="fast")
    cbs = [pipeline.cluster_single(ctx, p, sb) for sb in sbs]
    left_o, left = obs[0], cbs[0]
    ncl = left_o.n_clusters()
    keys, offs, post = left_o.index()
    cells = np.array([api.host_err_cell(e) for e in left.rep_view["hpc_err"]], np.uint8)
    ctx.left_load(ncl, cells, keys, offs, post)
    for c, src in [(0, ncl - 1), (2, 1), (ncl - 1, 0)]:
        old, new = _rep_values(left_o, c), _rep_values(left_o, src)
        left_o.update_mindb(c, old, new)
        ctx.index_update(c, old, new)
    left_o.cluster(right=obs[1], mode="fast")
    # the product: right representatives against the RESIDENT (updated) left state
    right = cbs[1]
    counts = np.bincount(right.member_cls, minlength=right.n_clusters).astype(np.int32)
    rv = dict(right.rep_view)
    rv.update(n_members=counts, depth=right.depth, min_cls_size=3)
    cls, strand, st = ctx.cluster_merge(p, dict(resident=True, cls_hpc_err=left.rep_view["hpc_err"]), rv)
    # oracle assignments of the right batch's reads after the merge
    ocl, ost = left_o.assignments(rs.n)
    rcl, rst = right.assignments(rs.n)
    reads = np.nonzero(rcl >= 0)[0]
    got = cls[rcl[reads]]
    got_s = strand[rcl[reads]].astype(np.int32) * rst[reads]
    assert np.array_equal(got, ocl[reads])
    assert np.array_equal(got_s, ost[reads])
    ctx.close()
