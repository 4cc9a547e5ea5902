#!/usr/bin/env python3
"""Developer driver for the alignment modes: cluster a synthetic batch in sahlin / furious mode on the
GPU (mapping + batched GPU aligner), print time and stats; --check compares with the oracle (slow: the
oracle aligns on one host core with its own scalar aligner)."""
import argparse
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import _lib, api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default=None)
ap.add_argument("--shape", default=None, help="n_reads,n_transcripts,length")
ap.add_argument("--mode", default="sahlin")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()

if a.shape:
    n, g, ln = (int(x) for x in a.shape.split(","))
    rs = synth.generate(n, g, ln, 10, 21, seed=a.seed)
else:
    rs = synth.generate_config(a.config or "config1", seed=a.seed)
B, view = oracle_sorted_batch(rs)
seqs = [rs.read(int(i))[0] for i in view["orig"]]
off = np.zeros(len(seqs) + 1, np.int64)
off[1:] = np.cumsum([len(s) for s in seqs])
v = dict(view)
v.update(raw_seq=b"".join(seqs), raw_off=off)
ctx = api.Context(0)
p = api.default_params(11, 15, a.mode)
for r in range(a.reps):
    t = time.time()
    cls, strand, st = ctx.cluster_batch(p, v)
    dt = time.time() - t
    print(f"rep {r}: {a.mode} {rs.tag}: {dt * 1e3:.1f} ms ({rs.n / dt:.0f} reads/s) {st}", flush=True)
if a.check:
    # (the oracle aligns with its own scalar aligner: nothing of the product behind it)
    t = time.time()
    ocl, ost, ostat = oracle_entry_assignments(B, view, mode=a.mode)
    dt = time.time() - t
    print(f"oracle {dt:.1f}s ({rs.n / dt:.1f} reads/s) {ostat}")
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    print("MISMATCHES", len(bad), bad[:10])
    sys.exit(1 if len(bad) else 0)
