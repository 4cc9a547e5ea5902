"""GPU parity of the merge path (`cluster -l L -r R`, src/cluster.cpp:67-322 with two real batches)
and of the exported MinDB (AddMinimizers, src/minimizer.cpp:31-42) against the oracle."""
import numpy as np
import pytest

from isonclust2_amd import api, pipeline, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _batches(rs, nb, k=11, w=15):
    """Globally score-sorted reads cut into nb consecutive batches (src/main.cpp:149-199)."""
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    p = po.default_params(k, w)
    cuts = np.linspace(0, rs.n, nb + 1).astype(int)
    obs, sbs = [], []
    for b in range(nb):
        B = po.Batch(R, int(cuts[b]), int(cuts[b + 1]) - 1, p, batch_nr=b)
        info, off_f, off_r, mn, ps = B.minimizer_soa()
        view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"],
                    hpc_len=info["hpc_len"], score=info["score"], raw_err=info["raw_err"],
                    hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8), min_qual=p.min_qual)
        obs.append(B)
        sbs.append(pipeline.SortedBatch(view=view, read_ids=info["orig"].astype(np.int64), batch_nr=b,
                                        batch_start=int(cuts[b]), batch_end=int(cuts[b + 1]) - 1))
    return obs, sbs


def _same_index(cb, B):
    keys, offs, post = B.index()
    assert np.array_equal(cb.mindb[0], keys)
    assert np.array_equal(cb.mindb[1], offs)
    assert np.array_equal(cb.mindb[2], post)


@pytest.mark.parametrize("cfg,seed", [("config1", 1), ("short_dup", 2)])
def test_index_export_equals_oracle_mindb(ctx, cfg, seed):
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, 1)
    obs[0].cluster(mode="fast")
    cb = pipeline.cluster_single(ctx, api.default_params(11, 15, "fast"), sbs[0])
    _same_index(cb, obs[0])
    ocl, ost = obs[0].assignments(rs.n)
    cls, strand = cb.assignments(rs.n)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)


@pytest.mark.parametrize("cfg,seed,nb", [("config1", 1, 2), ("config1", 2, 4), ("short_dup", 1, 3)])
def test_left_fold_merge_equals_oracle(ctx, cfg, seed, nb):
    """((b0 + b1) + b2) + ... as in the reference README's example fold."""
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, nb)
    p = api.default_params(11, 15, "fast")
    for B in obs:
        B.cluster(mode="fast")
    cbs = [pipeline.cluster_single(ctx, p, sb) for sb in sbs]
    for cb, B in zip(cbs, obs):
        _same_index(cb, B)
    left_o, left = obs[0], cbs[0]
    for b in range(1, nb):
        left_o.cluster(right=obs[b], mode="fast")
        left = pipeline.cluster_merge(ctx, p, left, cbs[b])
        ocl, ost = left_o.assignments(rs.n)
        cls, strand = left.assignments(rs.n)
        bad = np.nonzero((cls != ocl) | (strand != ost))[0]
        assert len(bad) == 0, (b, len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
        assert left.n_clusters == left_o.n_clusters()
        _same_index(left, left_o)
