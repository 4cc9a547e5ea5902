#!/usr/bin/env python3
"""Consensus mode on the first M sorted entries of a fuzz case, product against oracle: n g L cmin cmax period seed M.  Prints the first
MinDB difference."""
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
from isonclust2_amd import _lib, api, synth  # noqa: E402
from tests.helpers import ToyGraphs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
n, g, ln, cmin, cmax, period, seed, M = (int(x) for x in sys.argv[1:9])
rs = synth.generate(n, g, ln, 11, 22, seed=seed, dup_every=0)
R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
R.score_sort(11, 15)
p = po.default_params(11, 15)
p.cons_max_size = cmax
B = po.Batch(R, 0, M - 1, p)
info, off_f, off_r, mn, ps = B.minimizer_soa()
view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"], hpc_len=info["hpc_len"], score=info["score"],
            raw_err=info["raw_err"], hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8), min_qual=p.min_qual, orig=info["orig"])
gs = ToyGraphs()
po.lib().orc_set_consensus(C.cast(C.pointer(gs.ops), C.c_void_p), cmin, period)
try:
    ost = B.cluster(mode="fast")
finally:
    po.lib().orc_set_consensus(None, 50, 500)
seqs = [rs.read(int(i))[0] for i in view["orig"]]
off = np.zeros(len(seqs) + 1, np.int64)
off[1:] = np.cumsum([len(x) for x in seqs])
v = dict(view)
v.update(raw_seq=b"".join(seqs), raw_off=off)
ctx = api.Context(0)
pg = ToyGraphs()
cargs = _lib.ConsensusArgs(cons_min_size=cmin, cons_max_size=cmax, cons_period=period, left_depth=-1, left_sizes=None)
cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, "fast"), None, v, cargs, pg.ops)
keys, offs, post = ctx.index_export()
okeys, ooffs, opost = B.index()
print("entries", M, "cons", st["n_cons_invoked"], ost["cons_invoked"], "logs equal", pg.log == gs.log, "keys equal", np.array_equal(keys, okeys))
pm = {int(k): tuple(post[offs[i]:offs[i + 1]].tolist()) for i, k in enumerate(keys)}
om = {int(k): tuple(opost[ooffs[i]:ooffs[i + 1]].tolist()) for i, k in enumerate(okeys)}
nd = 0
for k in sorted(set(pm) | set(om)):
    if pm.get(k) != om.get(k):
        nd += 1
        if nd <= 8:
            print("key", k, "product", pm.get(k), "oracle", om.get(k))
print("differing keys", nd)
