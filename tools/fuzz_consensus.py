#!/usr/bin/env python3
"""Randomised differential run of consensus mode (toy graph store on both sides) against the oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import _lib, api, synth  # noqa: E402
from tests.helpers import ToyGraphs  # noqa: E402
from tests.test_consensus import _oracle_run  # noqa: E402

import ctypes as C  # noqa: E402

from oracle import pyoracle as po  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
mode = sys.argv[3] if len(sys.argv) > 3 else "fast"
hook = None   # (sahlin: the oracle's own scalar aligner; no product code behind the oracle)
ctx = api.Context(0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.integers(2, 220))
    g = int(rng.integers(1, 12))
    ln = int(rng.choice([300, 500, 800, 1200]))
    cmax = int(rng.choice([3, 6, 12, 50]))
    cmin = int(rng.choice([2, 3, 5, 20]))
    period = int(rng.choice([5, 25, 500]))
    seed = int(rng.integers(0, 1 << 30))
    if mode != "fast":
        n, ln = min(n, 120), min(ln, 800)
    rs = synth.generate(n, g, ln, 11, 22, seed=seed, dup_every=int(rng.choice([0, 0, 2])))
    tag = f"case {case}: n={n} g={g} L={ln} cons=({cmin},{cmax},{period}) seed={seed}"
    try:
        if hook is not None:
            po.lib().orc_set_aligner(C.cast(hook, C.c_void_p))
        try:
            B, view, ost, og = _oracle_run(rs, cmax, cmin, period, mode=mode)
        finally:
            po.lib().orc_set_aligner(None)
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
        ok = (np.array_equal(cls, ocl) and np.array_equal(strand, ostr) and pg.log == og.log and
              st["n_cons_invoked"] == ost["cons_invoked"] and np.array_equal(keys, okeys) and np.array_equal(offs, ooffs) and
              np.array_equal(post, opost))
        if not ok:
            bad += 1
            print("MISMATCH", tag, st["n_cons_invoked"], ost["cons_invoked"], flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(e), flush=True)
print(f"fuzz consensus: {n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
