#!/usr/bin/env python3
"""Debug aid: product leaf b of tests/config5_common.py against an oracle dump (tests/golden/_debug/leaf<b>*.{npz,json})."""
import json, sys, os
sys.path.insert(0, ".")
import numpy as np
from isonclust2_amd import api, pipeline
from tests import config5_common as c5
from tests.helpers import ToyGraphs
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = api.Context(0)
pre = sys.argv[2] if len(sys.argv) > 2 else ""
if pre:
    # the reduced tree (or part of it) first, in the SAME context
    rs1 = c5.reads(4, 1500)
    s1, _ = pipeline.sort_stage(ctx, rs1, c5.K, c5.W)
    p1 = api.default_params(c5.K, c5.W, c5.MODE)
    cbs, gs = [], []
    for bb in range(4 if pre != "leaf" else 1):
        st1 = ToyGraphs()
        cbs.append(pipeline.cluster_consensus_single(ctx, p1, pipeline.slice_sorted(s1, bb * 1500, (bb + 1) * 1500, batch_nr=bb), (c5.CONS_MIN, c5.CONS_MAX, c5.CONS_PERIOD), st1))
        gs.append(st1.g[0])
    if pre == "tree":
        for li, ri in c5.TREE:
            st1 = ToyGraphs()
            st1.g[0], st1.g[1] = gs[li], gs[ri]
            cbs[li] = pipeline.cluster_consensus_merge(ctx, p1, cbs[li], cbs[ri], (c5.CONS_MIN, c5.CONS_MAX, c5.CONS_PERIOD), st1)
            gs[li] = st1.g[0]
    print("pre-run done:", pre, flush=True)
rs = c5.reads()
srt, _ = pipeline.sort_stage(ctx, rs, c5.K, c5.W)
sb = pipeline.slice_sorted(srt, b * c5.PER, (b + 1) * c5.PER, batch_nr=b)
store = ToyGraphs()
p = api.default_params(c5.K, c5.W, c5.MODE)
from isonclust2_amd import _lib
cargs = _lib.ConsensusArgs(cons_min_size=c5.CONS_MIN, cons_max_size=c5.CONS_MAX, cons_period=c5.CONS_PERIOD, left_depth=-1, left_sizes=None)
cls, strand, st = ctx.cluster_consensus(p, None, sb.view, cargs, store.ops)
print(st)
d = np.load(f"tests/golden/_debug/leaf{b}.npz")
olog = [tuple(x) for x in json.load(open(f"tests/golden/_debug/leaf{b}_log.json"))]
assert np.array_equal(d["orig"], sb.read_ids), "sort order differs"
bad = np.nonzero((cls != d["cls"]) | (strand != d["strand"]))[0]
print("entries differing:", len(bad), bad[:10], cls[bad[:10]], d["cls"][bad[:10]], strand[bad[:10]], d["strand"][bad[:10]])
first = next((x for x in range(min(len(store.log), len(olog))) if store.log[x] != olog[x]), None)
print("log lengths", len(store.log), len(olog), "first differing op", first)
if first is not None:
    print("product:", store.log[first - 3:first + 4])
    print("oracle: ", olog[first - 3:first + 4])
