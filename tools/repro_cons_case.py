#!/usr/bin/env python3
"""One case of tools/fuzz_consensus.py by its parameters: n g L cmin cmax period seed [mode].  Prints what differs from the oracle."""
import sys

import numpy as np

import os
sys.path.insert(0, os.environ.get("IOC_TREE", "."))
from isonclust2_amd import _lib, api, synth  # noqa: E402
from tests.helpers import ToyGraphs  # noqa: E402
from tests.test_consensus import _oracle_run  # noqa: E402

n, g, ln, cmin, cmax, period, seed = (int(x) for x in sys.argv[1:8])
mode = sys.argv[8] if len(sys.argv) > 8 else "fast"
dup = int(sys.argv[9]) if len(sys.argv) > 9 else 0
ctx = api.Context(0)
rs = synth.generate(n, g, ln, 11, 22, seed=seed, dup_every=dup)
B, view, ost, og = _oracle_run(rs, cmax, cmin, period, mode=mode)
acl, ast = B.assignments(rs.n)
ocl, ostr = acl[view["orig"]], ast[view["orig"]]
seqs = [rs.read(int(i))[0] for i in view["orig"]]
off = np.zeros(len(seqs) + 1, np.int64)
off[1:] = np.cumsum([len(x) for x in seqs])
v = dict(view)
v.update(raw_seq=b"".join(seqs), raw_off=off)
pg = ToyGraphs()
cargs = _lib.ConsensusArgs(cons_min_size=cmin, cons_max_size=cmax, cons_period=period, left_depth=-1, left_sizes=None)
cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, mode), None, v, cargs, pg.ops)
keys, offs, post = ctx.index_export()
okeys, ooffs, opost = B.index()
d = np.nonzero(cls != ocl)[0]
print("cons_invoked", st["n_cons_invoked"], ost["cons_invoked"], "| first differing assignment", (int(d[0]), int(cls[d[0]]), int(ocl[d[0]])) if len(d) else None,
      "| strands equal", np.array_equal(strand, ostr), "| logs equal", pg.log == og.log, "| MinDB equal",
      np.array_equal(keys, okeys) and np.array_equal(offs, ooffs) and np.array_equal(post, opost))
for i, (a, b) in enumerate(zip(pg.log, og.log)):
    if a != b:
        print("first differing graph operation", i, a[:4] if isinstance(a, (list, tuple)) else a, "|", b[:4] if isinstance(b, (list, tuple)) else b)
        break
