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
