#!/usr/bin/env python3
"""Regenerates tests/golden/config5.json: the ORACLE's results on the workload of tests/config5_common.py (BASELINE.json
configs[4] at batch scale: 31 250-read x 2 kb batches, sahlin mode, consensus on, a 4-batch binary-tree merge).  Run in
the build container (one core, the oracle's own scalar aligner behind sahlin's fallback):

    python tools/gen_golden_config5.py [--per 31250] [--nb 4]

Keys of the file: "per<PER>" -> {"leaves": [...], "merges": [...]}; every record carries the FNV-1a digest of the
assignments of ALL nb*per reads, the cluster count, CONS_INVOKED / ALN_INVOKED, the sha of the graph-operation log and
of the MinDB, and the oracle's seconds."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as po  # noqa: E402
from tests import config5_common as c5  # noqa: E402
from tests.helpers import ToyGraphs, fnv1a  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--per", type=int, default=c5.PER)
ap.add_argument("--nb", type=int, default=c5.NB)
ap.add_argument("--real-graphs", action="store_true", help="the record `real<PER>`: REAL_NB leaves + their merge with the oracle's own POA behind the consensus hook "
                                                            "(tests/config5_common.py, VERDICT r4 item 4)")
a = ap.parse_args()
if a.real_graphs:
    import subprocess
    sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_golden_config5_real.py")] + (["--per", str(a.per)] if a.per != c5.PER else [])))
PATH = os.path.join(ROOT, "tests", "golden", "config5.json")
out = json.load(open(PATH)) if os.path.exists(PATH) else {}
rec = {"workload": f"{a.nb} x {a.per} reads of {c5.LEN} b, G={c5.G}, chunk seeds 1000.., transcript seed {c5.TR_SEED}, global sort; "
                   f"{c5.MODE} k={c5.K} w={c5.W}; ConsMinSize {c5.CONS_MIN} ConsMaxSize {c5.CONS_MAX} ConsPeriod {c5.CONS_PERIOD}; ToyGraphs",
       "leaves": [], "merges": []}
t00 = time.time()


def log(*x):
    print(f"[{time.time() - t00:7.1f} s]", *x, flush=True)


rs = c5.reads(a.nb, a.per)
n_total = rs.n
R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
R.score_sort(c5.K, c5.W)
p = po.default_params(c5.K, c5.W)
p.cons_max_size = c5.CONS_MAX
log("reads sorted", n_total)


def record(B, g, st, dt):
    acl, ast = B.assignments(n_total)
    keys, offs, post = B.index()
    return {"clusters": B.n_clusters(), "assigned": int(np.count_nonzero(acl >= 0)), "fnv1a": f"{fnv1a(acl, ast):016x}",
            "cons_invoked": st["cons_invoked"], "aln_invoked": st["aln_invoked"], "joins": st["joins"],
            "new_clusters": st["new_clusters"], "graph_ops": len(g.log), "log_sha": c5.log_sha(g.log),
            "mindb_keys": int(len(keys)), "mindb_postings": int(len(post)), "mindb_sha": c5.mindb_sha(keys, offs, post),
            "oracle_seconds_1core": round(dt, 1)}


batches, graphs = [], []
for b in range(a.nb):
    B = po.Batch(R, b * a.per, (b + 1) * a.per - 1, p, batch_nr=b)
    g = ToyGraphs()
    po.lib().orc_set_consensus(C.cast(C.pointer(g.ops), C.c_void_p), c5.CONS_MIN, c5.CONS_PERIOD)
    t0 = time.time()
    try:
        st = B.cluster(mode=c5.MODE)
    finally:
        po.lib().orc_set_consensus(None, 50, 500)
    r = record(B, g, st, time.time() - t0)
    r["batch"] = b
    rec["leaves"].append(r)
    log("leaf", r)
    batches.append(B)
    graphs.append(g.g[0])
    out[f"per{a.per}"] = rec
    json.dump(out, open(PATH, "w"), indent=1, sort_keys=True)

tree = [t for t in c5.TREE if t[0] < a.nb and t[1] < a.nb]
for li, ri in tree:
    g = ToyGraphs()
    g.g[0], g.g[1] = graphs[li], graphs[ri]
    po.lib().orc_set_consensus(C.cast(C.pointer(g.ops), C.c_void_p), c5.CONS_MIN, c5.CONS_PERIOD)
    t0 = time.time()
    try:
        st = batches[li].cluster(right=batches[ri], mode=c5.MODE)
    finally:
        po.lib().orc_set_consensus(None, 50, 500)
    r = record(batches[li], g, st, time.time() - t0)
    r["left"], r["right"] = li, ri
    rec["merges"].append(r)
    log("merge", r)
    graphs[li] = g.g[0]
    graphs[ri] = batches[ri] = None
    out[f"per{a.per}"] = rec
    json.dump(out, open(PATH, "w"), indent=1, sort_keys=True)
log("done")
