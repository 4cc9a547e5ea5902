"""SURVEY §8(c) golden item (3): the intermediate tables of the hot path, device against oracle, directly —
GetMinimizerHits + ConsolidateMinimizerHits + SortMinimizerHits (src/minimizer.cpp:44-76, src/cluster.cpp:609-636):
per query the multiset of (cluster, strand, Size, Index of the first hit) over the clusters existing when the loop
reaches it; getMappedRatio (src/cluster.cpp:324-353): totalMapped of every candidate the reference's walk evaluates
(and of every candidate the device evaluated on top of that)."""
import numpy as np
import pytest

from isonclust2_amd import api, synth
from oracle import pyoracle as po
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _compare(ctx, view, rows, calls, entries, tgt):
    n = len(tgt)
    opener = tgt < 0
    # single batch, L = 0: target = entry that opened the cluster -> cluster id in creation order
    cid = np.full(n, -1, np.int64)
    gated = np.asarray(view["state"]) != 0
    cid[opener & ~gated] = np.arange(int((opener & ~gated).sum()))
    n_rows = n_walked = n_dev_eval = n_bound = 0
    for e in entries:
        m = rows["entry"] == e
        t, s, sz, fi, tm = ctx.query_candidates(int(e), 2 * n + 2)
        dev = sorted(zip(cid[t].tolist(), s.tolist(), sz.tolist(), fi.tolist()))
        orc = sorted(zip(rows["cls"][m].tolist(), rows["strand"][m].tolist(), rows["size"][m].tolist(),
                         rows["first_index"][m].tolist()))
        assert dev == orc, (e, len(dev), len(orc))
        n_rows += len(orc)
        tot = {(c, st): (x, w) for c, st, x, w in zip(rows["cls"][m].tolist(), rows["strand"][m].tolist(),
                                                      rows["total_mapped"][m].tolist(), rows["walked"][m].tolist())}
        need = api.host_min_total(int(view["hpc_len"][e]), 0.65)
        for c, st, x in zip(cid[t].tolist(), s.tolist(), tm.tolist()):
            want, walked = tot[(c, st)]
            if x == 0xFFFFFFFE:
                # rejected by the upper bound of totalMapped (k_gap_bounds) without an evaluation: the bound is sound iff the
                # oracle's exact total fails the threshold as well
                assert want < need, (e, c, st, want, need)
                n_bound += 1
                n_walked += 1 if walked else 0
                continue
            if walked:
                assert x == want, (e, c, st, x, want)          # the reference called getMappedRatio here
                n_walked += 1
            if x != 0xFFFFFFFF:
                assert x == want, (e, c, st, x, want)          # whatever the device evaluated is the oracle's value
                n_dev_eval += 1
    return n_rows, n_walked, n_dev_eval + n_bound


@pytest.mark.parametrize("cfg,seed,step", [("config1", 1, 7), ("short_dup", 1, 1), ("tiny", 7, 1)])
def test_candidate_tables_and_mapped_totals(ctx, cfg, seed, step):
    rs = synth.generate_config(cfg, seed=seed)
    B, view = oracle_sorted_batch(rs)
    n = rs.n
    entries = list(range(0, n, step))
    assert len(entries) >= 10
    po.trace_set(entries, mapped_calls=True)
    try:
        ocl, ost, ostat = oracle_entry_assignments(B, view)
        rows, calls = po.trace_rows(), po.trace_mapped_calls()
    finally:
        po.trace_set(())
    p = api.default_params(11, 15, "fast")
    cls, strand, st = ctx.cluster_batch(p, view)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
    tgt, _, _ = ctx.decisions()
    n_rows, n_walked, n_dev = _compare(ctx, view, rows, calls, entries, tgt)
    assert n_rows == len(rows["entry"]) and n_rows > 10 * len(entries) // 4
    # every getMappedRatio call of the traced entries was seen
    assert n_walked == int(np.isin(calls["entry"], entries).sum())
    assert n_dev >= n_walked


def test_bound_of_total_mapped_is_a_pure_shortcut(ctx, monkeypatch):
    """The upper bound of totalMapped (k_gap_bounds) and the cut of the candidate lists change no decision: with them, without
    the list cut, without both — on a tie-prone read set, a tiny one and the CPU-runnable configuration."""
    for cfg, seed in (("short_dup", 3), ("tiny", 5), ("config1", 2)):
        rs = synth.generate_config(cfg, seed=seed)
        _, view = oracle_sorted_batch(rs)
        p = api.default_params(11, 15, "fast")
        res = []
        for bound, keepq in (("1", "1"), ("1", "0"), ("0", "0")):
            monkeypatch.setenv("IOC_RESOLVE_BOUND", bound)
            monkeypatch.setenv("IOC_SCORE_KEEPQ", keepq)
            cls, strand, st = ctx.cluster_batch(p, view)
            res.append((cls.copy(), strand.copy(), int(st["n_clusters"]), int(ctx.timings()["n_mapped_evals"])))
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2], cfg
        assert res[0][3] <= res[1][3] <= res[2][3]          # each half only ever removes evaluations
