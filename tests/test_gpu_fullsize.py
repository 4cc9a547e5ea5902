"""BASELINE.json configs[1..3] at FULL size on the GPU against committed oracle goldens
(tests/golden/oracle_assignments.json, generated in the build container by tools/gen_golden.py --full /
--full-sahlin: FNV-1a digests of the oracle's assignments, cluster counts, alignment-fallback counts).

  configs[1]  3000 reads / 50 Mb, k=11 w=15, fast mode               test_config2_fast_full
  configs[2]  the same batch, sahlin mode (GPU alignment fallback)   test_config2_sahlin_full
  configs[3]  8 batches x 3000 reads (seeds 1..8), each clustered on its own, folded left to right
              (README.md:105-117; src/cluster.cpp:67-322 with two batches)   test_config4_fold_*

Everything runs through the C ABI from host arrays (the *core* region of SURVEY §8(d)): GPU sort stage ->
ioc_cluster_merge -> assignments + ioc_index_export.  The oracle is not run here (minutes per batch on a core; in
sahlin mode a quarter of an hour): the goldens stand for it."""
import json
import os

import numpy as np
import pytest

from isonclust2_amd import api, pipeline, synth
from tests.helpers import fnv1a

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_assignments.json")))
K, W = 11, 15


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def batches(ctx):
    """The sorted batches of seeds 1..8, prepared once by the product's own GPU sort stage."""
    cache = {}

    def get(seed):
        if seed not in cache:
            rs = synth.generate_config("config2", seed=seed)
            sb, order = pipeline.sort_stage(ctx, rs, K, W, read_id_base=rs.n * (seed - 1), batch_nr=seed - 1)
            cache[seed] = (sb, rs.n)
        return cache[seed]
    return get


def _single(ctx, batches, seed, mode):
    sb, n = batches(seed)
    cb = pipeline.cluster_single(ctx, api.default_params(K, W, mode), sb)
    acl, ast = cb.assignments(sb.batch_start + n)
    cls_e, strand_e = acl[sb.read_ids], ast[sb.read_ids]   # per entry of the sorted batch
    return cb, cls_e, strand_e


def test_config2_fast_full(ctx, batches):
    g = GOLD["config2:1"]
    cb, cls, strand = _single(ctx, batches, 1, "fast")
    assert cb.n_clusters == g["clusters"]
    assert f"{fnv1a(cls, strand):016x}" == g["fnv1a"]
    # the exported MinDB holds exactly the postings AddMinimizers appended (src/minimizer.cpp:31-42)
    assert len(cb.mindb[2]) == g["stats"]["index_appends"]
    # the raw hits GetMinimizerHits emits on this batch (src/minimizer.cpp:44-76), recounted on the device
    assert ctx.count_reference_postings() == g["stats"]["postings"]


def test_config2_other_array_layouts_and_no_upload_overlap(ctx, batches, monkeypatch):
    """ioc_cluster_merge sends the forward lists' values first and the rest from a copy thread when the arrays are laid out
    [all forward][all reverse]; any other layout (here: reverse block first), and IOC_UPLOAD_OVERLAP=0, give the same result."""
    import dataclasses
    g = GOLD["config2:1"]
    sb, n = batches(1)
    v = sb.view
    of, orv = np.asarray(v["off_fwd"]), np.asarray(v["off_rev"])
    assert of[0] == 0 and orv[0] == of[-1]                 # the product's own layout
    nf, nr = int(of[-1]), int(orv[-1] - orv[0])
    v2 = dict(v)
    v2["min_val"] = np.concatenate([v["min_val"][nf:nf + nr], v["min_val"][:nf]])
    v2["min_pos"] = np.concatenate([v["min_pos"][nf:nf + nr], v["min_pos"][:nf]])
    v2["off_rev"] = orv - nf
    v2["off_fwd"] = of + nr
    sb2 = dataclasses.replace(sb, view=v2)
    p = api.default_params(K, W, "fast")
    for batch, env in ((sb2, None), (sb, "0")):
        if env is not None:
            monkeypatch.setenv("IOC_UPLOAD_OVERLAP", env)
        cb = pipeline.cluster_single(ctx, p, batch)
        acl, ast = cb.assignments(sb.batch_start + n)
        assert cb.n_clusters == g["clusters"]
        assert f"{fnv1a(acl[sb.read_ids], ast[sb.read_ids]):016x}" == g["fnv1a"]
        assert len(cb.mindb[2]) == g["stats"]["index_appends"]


def test_config2_sahlin_full(ctx, batches):
    g = GOLD["config2:1:sahlin"]
    cb, cls, strand = _single(ctx, batches, 1, "sahlin")
    assert cb.n_clusters == g["clusters"]
    assert cb.stats["n_aln_invoked"] == g["stats"]["aln_invoked"]      # reads reaching getBestClusterAln (cluster.cpp:563)
    assert f"{fnv1a(cls, strand):016x}" == g["fnv1a"]
    assert len(cb.mindb[2]) == g["stats"]["index_appends"]


@pytest.mark.parametrize("mode", ["fast", "sahlin"])
def test_config4_fold(ctx, batches, mode):
    """configs[3] on one GPU: every batch clustered on its own (checked against its golden), then
    ((b1 + b2) + b3) ... with the merge path; cluster counts after every merge and the digest over all 24 000 reads."""
    import torch
    from isonclust2_amd import dist as idist
    g = GOLD[f"config4:{mode}"]
    dev = torch.device("cuda", 0)
    cbs, parts, metas = [], [], []
    for seed in g["seeds"]:
        cb, cls, strand = _single(ctx, batches, seed, mode)
        gs = GOLD[f"config2:{seed}" + ("" if mode == "fast" else ":sahlin")]
        assert cb.n_clusters == gs["clusters"], seed
        assert f"{fnv1a(cls, strand):016x}" == gs["fnv1a"], seed
        cbs.append(cb)
        # what this batch's rank contributes to the merge's all-gather: its representatives' lists, gathered on the device
        parts.append(idist.gather_local(ctx, cb, torch, dev))
        metas.append(idist.unpack_clustered(idist.pack_clustered(cb, with_minimizers=False)))
    p = api.default_params(K, W, mode)
    # the merge as the ranks run it (device-resident records, ONE pass): digest over all 24 000 reads
    cap = max(int(m.numel()) for m, _ in parts)
    pad = lambda t: torch.cat([t, torch.zeros(cap - int(t.numel()), dtype=torch.int32, device=dev)])
    one = idist.merge_gathered(ctx, p, metas, torch.cat([pad(m) for m, _ in parts]), torch.cat([pad(q) for _, q in parts]), cap)
    del parts
    ocl, ost = one.assignments(g["n"])
    assert one.n_clusters == g["clusters"] and f"{fnv1a(ocl, ost):016x}" == g["fnv1a"]
    if mode == "sahlin":
        assert one.stats["n_aln_invoked"] == sum(st["stats"]["aln_invoked"] for st in g["steps"])
    # and the reference's own order of work: one merge after the other
    left = cbs[0]
    for step, cb in zip(g["steps"], cbs[1:]):
        left = pipeline.cluster_merge(ctx, p, left, cb)
        assert left.n_clusters == step["clusters_after"], step["right_seed"]
        if mode == "sahlin":
            assert left.stats["n_aln_invoked"] == step["stats"]["aln_invoked"], step["right_seed"]
    acl, ast = left.assignments(g["n"])
    assert int(np.count_nonzero(acl >= 0)) == g["assigned"]
    assert f"{fnv1a(acl, ast):016x}" == g["fnv1a"]
