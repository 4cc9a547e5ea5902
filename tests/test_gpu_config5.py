"""BASELINE.json configs[4] ("config 5": 2 M reads / 64 batches of 31 250 x 2 kb, sahlin mode, consensus on) at batch
scale on the GPU against committed ORACLE goldens (tests/golden/config5.json, tools/gen_golden_config5.py; workload in
tests/config5_common.py): one 31 250-read batch with the consensus branch of ClusterSortedReads (src/cluster.cpp:263-309,
src/consensus.cpp:34-126, UpdateMinDB src/minimizer.cpp:124-160), and the 4-batch binary-tree merge of such batches
(src/cluster.cpp:67-322 with two batches, consensus on: ConsMinSize 2, right graphs' sizes as weights).

Compared per step: FNV-1a digest of the assignments of all reads, cluster count, CONS_INVOKED, ALN_INVOKED, the sha of
the graph-operation log (the exact sequence of create / add / consensus / purge calls with their lengths and weights)
and the sha of the final MinDB.  Everything goes through the C ABI (ioc_cluster_consensus) from host arrays; the graphs
are the ToyGraphs store on both sides (spoa is absent from the reference tree: graph parity is unpinned).

A reduced copy of the same tree (4 x 1500 reads) runs first: seconds, and it localises a failure."""
import json
import os

import numpy as np
import pytest

from isonclust2_amd import api, pipeline
from tests import config5_common as c5
from tests.helpers import ToyGraphs, fnv1a

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5.json")))
CONS = (c5.CONS_MIN, c5.CONS_MAX, c5.CONS_PERIOD)


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _check(tag, cb, store, g, n_total):
    acl, ast = cb.assignments(n_total)
    got = {"clusters": cb.n_clusters, "assigned": int(np.count_nonzero(acl >= 0)), "fnv1a": f"{fnv1a(acl, ast):016x}",
           "cons_invoked": int(cb.stats["n_cons_invoked"]), "aln_invoked": int(cb.stats["n_aln_invoked"]),
           "graph_ops": len(store.log), "log_sha": c5.log_sha(store.log), "mindb_keys": int(len(cb.mindb[0])),
           "mindb_postings": int(len(cb.mindb[2])), "mindb_sha": c5.mindb_sha(*cb.mindb)}
    want = {k: g[k] for k in got}
    assert got == want, (tag, {k: (got[k], want[k]) for k in got if got[k] != want[k]})


class Tree:
    """The leaves and merges of one golden record, computed on demand and kept (the leaf test and the tree test share them)."""

    def __init__(self, ctx, per):
        self.ctx, self.per, self.gold = ctx, per, GOLD[f"per{per}"]
        self.nb = c5.NB          # the batches are cuts of ONE global sort over all nb * per reads: nb is part of the workload
        assert len(self.gold["leaves"]) == self.nb and len(self.gold["merges"]) == len(c5.TREE), "golden record incomplete"
        self.n_total = self.nb * per
        self.p = api.default_params(c5.K, c5.W, c5.MODE)
        self.sorted = None
        self.cb, self.graphs = {}, {}

    def leaf(self, b):
        if b not in self.cb:
            if self.sorted is None:
                rs = c5.reads(self.nb, self.per)
                self.sorted, _ = pipeline.sort_stage(self.ctx, rs, c5.K, c5.W)      # the product's own GPU sort stage, global order
            sb = pipeline.slice_sorted(self.sorted, b * self.per, (b + 1) * self.per, batch_nr=b)
            store = ToyGraphs()
            cb = pipeline.cluster_consensus_single(self.ctx, self.p, sb, CONS, store)
            _check(f"leaf {b}", cb, store, self.gold["leaves"][b], self.n_total)
            self.cb[b], self.graphs[b] = cb, store.g[0]
        return self.cb[b]

    def merge(self, step):
        g = self.gold["merges"][step]
        li, ri = g["left"], g["right"]
        store = ToyGraphs()
        store.g[0], store.g[1] = self.graphs[li], self.graphs[ri]
        cb = pipeline.cluster_consensus_merge(self.ctx, self.p, self.cb[li], self.cb[ri], CONS, store)
        _check(f"merge {li}+{ri}", cb, store, g, self.n_total)
        self.cb[li], self.graphs[li] = cb, store.g[0]
        del self.cb[ri], self.graphs[ri]
        return cb


@pytest.fixture(scope="module")
def small(ctx):
    return Tree(ctx, 1500)


@pytest.fixture(scope="module")
def full(ctx):
    return Tree(ctx, c5.PER)


def test_config5_reduced_tree(small, monkeypatch):
    monkeypatch.setenv("IOC_CONS_VIEW_CHECK", "1")   # the left view patched across passes must equal a rebuilt one
    for b in range(small.nb):
        small.leaf(b)
    for s in range(len(small.gold["merges"])):
        small.merge(s)
    assert small.gold["merges"][0]["cons_invoked"] > 100      # the merges really take consensus events


def test_config5_batch_consensus(full):
    """ONE 31 250 x 2 kb batch, sahlin, consensus on."""
    cb = full.leaf(0)
    g = full.gold["leaves"][0]
    assert g["cons_invoked"] > 1000 and cb.stats["n_cons_invoked"] == g["cons_invoked"]


def test_config5_tree_merge(full):
    """The other three leaves and the binary tree (b0 + b1), (b2 + b3), ((b0 + b1) + (b2 + b3))."""
    for b in range(full.nb):
        full.leaf(b)
    for s in range(len(full.gold["merges"])):
        cb = full.merge(s)
    assert cb.batch_start == 0 and cb.batch_end == full.n_total - 1


# ---- REAL partial-order graphs: the product's engine against the oracle's POA goldens -----------------------
class PoaStore:
    """The product's POA engine (ioc_poa.hip) as the graph store of ioc_cluster_consensus, with what the pipeline needs around
    it: the replaced representatives' records (rep_changed: the consensus strings arrive there in event order) and the transfer of
    graphs between engines (ioc_poa_graph_save / load: what the `.cer` files carry from a leaf to a merge)."""

    def __init__(self, ctx):
        import ctypes as C
        from isonclust2_amd import _lib
        from tests.test_gpu_poa import Poa
        self.C, self.poa = C, Poa(ctx)
        self.rep_records, self.events = [], []

        def rep_changed(user, cls, rec):
            r = rec.contents
            take = lambda ptr, n: np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            raw = C.string_at(r.raw_seq, r.raw_len)
            self.events.append((int(cls), raw))
            self.rep_records.append((int(cls), dict(raw_seq=raw, raw_len=int(r.raw_len), raw_err=float(r.raw_err), score=float(r.raw_score),
                                                    hpc_len=int(r.hpc_len), hpc_err=float(r.hpc_err), fwd_min=take(r.fwd_min, r.n_fwd),
                                                    fwd_pos=take(r.fwd_pos, r.n_fwd), rev_min=take(r.rev_min, r.n_rev), rev_pos=take(r.rev_pos, r.n_rev),
                                                    entry=int(r.entry))))

        self._keep = _lib.CONS_REP_CHANGED(rep_changed)
        self.ops = self.poa.ops
        self.ops.rep_changed = self._keep

    def graph(self, c, side=0):
        return self.poa.graph(c, side)

    def consensus(self, c, side=0):
        return self.poa.consensus(c, side)

    def copy_graph_to(self, idx, other, to_side, to_idx):
        C, L = self.C, self.poa.L
        L.ioc_poa_graph_save.restype = C.c_int64
        n = L.ioc_poa_graph_save(self.poa.h, 0, idx, None, C.c_int64(0))
        assert n > 0
        buf = (C.c_uint8 * n)()
        assert L.ioc_poa_graph_save(self.poa.h, 0, idx, buf, C.c_int64(n)) == n
        assert L.ioc_poa_graph_load(other.poa.h, to_side, to_idx, buf, C.c_int64(n)) == 0

    def close(self):
        self.poa.close()


def _check_real(tag, cb, store, g, n_total):
    acl, ast = cb.assignments(n_total)
    gs, cs = c5.graphs_sha(store, cb.n_clusters)
    got = {"clusters": cb.n_clusters, "assigned": int(np.count_nonzero(acl >= 0)), "fnv1a": f"{fnv1a(acl, ast):016x}",
           "cons_invoked": int(cb.stats["n_cons_invoked"]), "aln_invoked": int(cb.stats["n_aln_invoked"]), "events": len(store.events),
           "events_sha": c5.events_sha(store.events), "graphs_sha": gs, "consensus_sha": cs, "mindb_keys": int(len(cb.mindb[0])),
           "mindb_postings": int(len(cb.mindb[2])), "mindb_sha": c5.mindb_sha(*cb.mindb)}
    want = {k: g[k] for k in got}
    assert got == want, (tag, {k: (got[k], want[k]) for k in got if got[k] != want[k]})


@pytest.mark.parametrize("per", [200, c5.REAL_PER])
def test_config5_real_graphs(ctx, per):
    """Two leaves of `per` reads x 2 kb (sahlin, -c 150, ConsMinSize 20) and their merge with the product's POA engine behind the
    consensus, against the goldens the ORACLE computed with its own scalar POA behind its hook (tools/gen_golden_config5.py
    --real-graphs): assignments, event counts, the consensus strings in event order, every cluster's final graph (letters,
    ranks, weighted edges) and consensus, the MinDB.  (src/consensus.cpp:34-126, src/cluster.cpp:263-309; spoa itself: unpinned.)"""
    gold = GOLD[f"real{per}"]
    nb = c5.REAL_NB
    rs = c5.real_reads(nb, per)
    p = api.default_params(c5.K, c5.W, c5.MODE)
    srt, _ = pipeline.sort_stage(ctx, rs, c5.K, c5.W)
    cbs, stores = [], []
    try:
        for b in range(nb):
            sb = pipeline.slice_sorted(srt, b * per, (b + 1) * per, batch_nr=b)
            store = PoaStore(ctx)
            stores.append(store)
            cb = pipeline.cluster_consensus_single(ctx, p, sb, CONS, store)
            _check_real(f"leaf {b}", cb, store, gold["leaves"][b], rs.n)
            cbs.append(cb)
        if per == c5.REAL_PER:
            assert gold["leaves"][0]["cons_invoked"] > 500      # the leaves really take consensus events
        gm = PoaStore(ctx)
        stores.append(gm)
        for src, side in ((stores[0], 0), (stores[1], 1)):
            for c_id in range(cbs[side].n_clusters):
                src.copy_graph_to(c_id, gm, side, c_id)
        merged = pipeline.cluster_consensus_merge(ctx, p, cbs[0], cbs[1], CONS, gm)
        _check_real("merge 0+1", merged, gm, gold["merges"][0], rs.n)
        assert gold["merges"][0]["cons_invoked"] > 20
    finally:
        for s in stores:
            s.close()
