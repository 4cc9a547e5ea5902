#!/usr/bin/env python3
"""The oracle's candidate table of one entry in a consensus-mode run: n g L cmin cmax period seed entry."""
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
from isonclust2_amd import synth  # noqa: E402
from tests.helpers import ToyGraphs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
n, g, ln, cmin, cmax, period, seed, entry = (int(x) for x in sys.argv[1:9])
rs = synth.generate(n, g, ln, 11, 22, seed=seed, dup_every=0)
R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
R.score_sort(11, 15)
p = po.default_params(11, 15)
p.cons_max_size = cmax
B = po.Batch(R, 0, rs.n - 1, p)
gs = ToyGraphs()
po.lib().orc_set_consensus(C.cast(C.pointer(gs.ops), C.c_void_p), cmin, period)
po.trace_set([entry], mapped_calls=True)
try:
    st = B.cluster(mode="fast")
finally:
    po.lib().orc_set_consensus(None, 50, 500)
rows = po.trace_rows()
print("entry info", {k: v[entry] for k, v in B.entry_info().items() if hasattr(v, "__len__")})
for i in range(len(rows["entry"])):
    print({k: int(v[i]) for k, v in rows.items()})
mc = po.trace_mapped_calls()
for i in range(len(mc["entry"])):
    if mc["entry"][i] == entry:
        print("mapped call", {k: (float(v[i]) if v.dtype.kind == "f" else int(v[i])) for k, v in mc.items()})
acl, ast = B.assignments(rs.n)
print("oracle assignment of the entry", acl[B.entry_info()["orig"][entry]] if "orig" in B.entry_info() else None)
# which graph operations happened on clusters 0 and 11 before this entry?
ops = [o for o in gs.log]
print("graph operations in total", len(ops), "; the 148th..150th", ops[146:151])
