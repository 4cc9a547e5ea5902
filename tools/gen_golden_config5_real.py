#!/usr/bin/env python3
"""tests/golden/config5.json, record `real<PER>`: the consensus branch with REAL partial-order graphs on the ORACLE's side —
oracle/poa_oracle.cpp behind the oracle's consensus hook — on the workload of tests/config5_common.py (REAL_NB leaves of REAL_PER
reads x 2 kb, sahlin, -c 150, ConsMinSize 20, and their merge).  Run once in the build container (CPU, minutes):

    python tools/gen_golden_config5.py --real-graphs        (or this file directly, [--per N])

Per step: the digests of the toy-graph record plus the consensus strings in event order, every cluster's graph and consensus."""
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
from tests.helpers import fnv1a  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--per", type=int, default=c5.REAL_PER)
a = ap.parse_args()
PATH = os.path.join(ROOT, "tests", "golden", "config5.json")
out = json.load(open(PATH)) if os.path.exists(PATH) else {}
NB = c5.REAL_NB
rec = {"workload": f"{NB} x {a.per} reads of {c5.LEN} b, G={c5.REAL_G}, chunk seeds 2000.., transcript seed {c5.TR_SEED + 1}, global sort; {c5.MODE} "
                   f"k={c5.K} w={c5.W}; ConsMinSize {c5.CONS_MIN} ConsMaxSize {c5.CONS_MAX} ConsPeriod {c5.CONS_PERIOD}; graphs: oracle/poa_oracle.cpp",
       "leaves": [], "merges": []}
t00 = time.time()


def log(*x):
    print(f"[{time.time() - t00:7.1f} s]", *x, flush=True)


class LoggedPoa:
    """An OraclePoa whose consensus operation also records (cluster, string) per call: the events in their order."""

    def __init__(self):
        self.poa = po.OraclePoa()
        self.events = []
        inner = self.poa._cons
        user = self.poa.ops.user

        def cons(_user, side, idx, out, cap):
            n = inner(user, side, idx, C.cast(out, C.c_char_p), cap)
            if n >= 0 and side == 0:
                self.events.append((idx, C.string_at(out, n)))
            return n

        self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int)(cons)
        self.ops = po.PoaOps(self.poa.ops.user, self.poa.ops.create, self.poa.ops.size, self.poa.ops.add, C.cast(self._cb, C.c_void_p).value, self.poa.ops.purge)

    def pointer(self):
        return C.cast(C.pointer(self.ops), C.c_void_p)


rs = c5.real_reads(NB, a.per)
n_total = rs.n
R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
R.score_sort(c5.K, c5.W)
p = po.default_params(c5.K, c5.W)
p.cons_max_size = c5.CONS_MAX
log("reads sorted", n_total)


def record(B, g, st, dt):
    acl, ast = B.assignments(n_total)
    keys, offs, post = B.index()
    gs, cs = c5.graphs_sha(g.poa, B.n_clusters())
    return {"clusters": B.n_clusters(), "assigned": int(np.count_nonzero(acl >= 0)), "fnv1a": f"{fnv1a(acl, ast):016x}",
            "cons_invoked": st["cons_invoked"], "aln_invoked": st["aln_invoked"], "joins": st["joins"], "new_clusters": st["new_clusters"],
            "events": len(g.events), "events_sha": c5.events_sha(g.events), "graphs_sha": gs, "consensus_sha": cs,
            "mindb_keys": int(len(keys)), "mindb_postings": int(len(post)), "mindb_sha": c5.mindb_sha(keys, offs, post),
            "oracle_seconds_1core": round(dt, 1)}


batches, stores = [], []
for b in range(NB):
    B = po.Batch(R, b * a.per, (b + 1) * a.per - 1, p, batch_nr=b)
    g = LoggedPoa()
    po.lib().orc_set_consensus(g.pointer(), c5.CONS_MIN, c5.CONS_PERIOD)
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
    stores.append(g)
    out[f"real{a.per}"] = rec
    json.dump(out, open(PATH, "w"), indent=1, sort_keys=True)

gm = LoggedPoa()
for src, side in ((stores[0], 0), (stores[1], 1)):
    for c_id in range(batches[0 if side == 0 else 1].n_clusters()):
        src.poa.copy_graph_to(c_id, gm.poa, side, c_id)
po.lib().orc_set_consensus(gm.pointer(), c5.CONS_MIN, c5.CONS_PERIOD)
t0 = time.time()
try:
    st = batches[0].cluster(right=batches[1], mode=c5.MODE)
finally:
    po.lib().orc_set_consensus(None, 50, 500)
r = record(batches[0], gm, st, time.time() - t0)
r["left"], r["right"] = 0, 1
rec["merges"].append(r)
log("merge", r)
out[f"real{a.per}"] = rec
json.dump(out, open(PATH, "w"), indent=1, sort_keys=True)
log("done")
