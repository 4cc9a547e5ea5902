"""GPU tests of the individual C-ABI entry points (the steps the host driver composes) and of running on
a caller-owned HIP stream."""
import numpy as np
import pytest

from isonclust2_amd import api, synth
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    # torch brings its own HIP runtime: when both live in one process torch has to initialise first
    # (bench.py does the same); the stream test below needs it
    import torch
    torch.zeros(1, device="cuda")
    c = api.Context(0)
    yield c
    c.close()


def _upload(ctx, view, p):
    ctx.set_params(p)
    n = len(view["hpc_len"])
    cell = np.array([api.host_err_cell(e) for e in view["hpc_err"]], np.uint8)
    need = np.array([api.host_min_total(h, p.mapped_threshold) for h in view["hpc_len"]], np.uint32)
    ctx.queries_upload(view["off_fwd"], view["off_rev"], view["min_val"], view["min_pos"], view["hpc_len"], cell, need)
    ctx.left_load(0, None, None, None, None)
    return n


def test_step_by_step_equals_driver_and_oracle(ctx):
    rs = synth.generate_config("config1", seed=8)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, ostat = oracle_entry_assignments(B, view)
    p = api.default_params(11, 15, "fast")
    n = _upload(ctx, view, p)
    ctx.index_build()
    ctx.score()
    sweeps = ctx.resolve()
    assert sweeps >= 1
    tgt, strand, flags = ctx.decisions()
    # decisions -> cluster ids in creation order
    cid = np.full(n, -1, np.int32)
    cid[tgt < 0] = np.arange(int((tgt < 0).sum()))
    cls = np.where(tgt < 0, cid, cid[np.maximum(tgt, 0)])
    assert (flags & 1).sum() == 0                      # no order-dependent ties in this data set
    assert np.array_equal(cls, ocl)
    assert np.array_equal(np.where(tgt < 0, 1, strand), ost)
    # the reference's postings count, recounted on the device
    assert ctx.count_reference_postings() == ostat["postings"]
    # candidate table of one joined query: its winner is in it, with a passing totalMapped
    j = int(np.nonzero(tgt >= 0)[0][-1])
    t, s, sz, fi, tm = ctx.query_candidates(j, 2 * n)
    k = np.nonzero((t == tgt[j]) & (s == strand[j]))[0]
    assert len(k) == 1 and tm[k[0]] != 0xFFFFFFFF and tm[k[0]] >= api.host_min_total(view["hpc_len"][j], 0.65)
    assert sz[k[0]] == sz[(t >= 0)].max() or tm[np.argmax(sz)] < api.host_min_total(view["hpc_len"][j], 0.65)
    # forcing that query to open a cluster changes later decisions consistently, clearing restores them
    ctx.force_decision(j, -1)
    ctx.resolve()
    t2, _, _ = ctx.decisions()
    assert t2[j] == -1 and np.array_equal(t2[:j], tgt[:j])
    ctx.L.ioc_clear_forced(ctx.h)
    ctx.resolve()
    t3, s3, _ = ctx.decisions()
    assert np.array_equal(t3, tgt) and np.array_equal(s3, strand)


def test_call_order_is_enforced(ctx):
    rs = synth.generate_config("tiny")
    B, view = oracle_sorted_batch(rs)
    _upload(ctx, view, api.default_params(11, 15, "fast"))
    with pytest.raises(api.IocError) as e:
        ctx.score()
    assert e.value.code == -3
    ctx.index_build()
    with pytest.raises(api.IocError):
        ctx.resolve()
    with pytest.raises(api.IocError):
        ctx.queries_upload(view["off_fwd"], view["off_rev"], view["min_val"], view["min_pos"], view["hpc_len"],
                           np.full(len(view["hpc_len"]), 16, np.uint8), np.zeros(len(view["hpc_len"]), np.uint32))


def test_runs_on_a_caller_owned_stream(ctx):
    import torch
    rs = synth.generate_config("config1", seed=2)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, _ = oracle_entry_assignments(B, view)
    st = torch.cuda.Stream()
    assert ctx.L.ioc_set_stream(ctx.h, st.cuda_stream) == 0
    try:
        cls, strand, _ = ctx.cluster_batch(api.default_params(11, 15, "fast"), view)
    finally:
        ctx.L.ioc_set_stream(ctx.h, None)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
