#!/usr/bin/env python3
"""Developer driver (not part of the product): prepare a synthetic sorted batch with the oracle,
run the HIP path on it, print timings, optionally compare with the oracle's clustering."""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config1")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--check", action="store_true")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

t = time.time()
rs = synth.generate_config(a.config, seed=a.seed)
print(f"generated {rs.tag} in {time.time() - t:.1f}s", flush=True)
t = time.time()
B, view = oracle_sorted_batch(rs)
print(f"oracle sort-stage prep {time.time() - t:.1f}s; minimizers {len(view['min_val'])}", flush=True)
ctx = api.Context(0)
p = api.default_params(11, 15, "fast")
for r in range(a.reps):
    t = time.time()
    cls, strand, st = ctx.cluster_batch(p, view)
    dt = time.time() - t
    print(f"rep {r}: cluster_batch {dt * 1e3:.1f} ms ({rs.n / dt:.0f} reads/s) stats={st} timings={ctx.timings()}",
          flush=True)
for r in range(a.reps):
    t = time.time()
    cls2, strand2, st2 = ctx.cluster_resident()
    dt = time.time() - t
    print(f"resident rep {r}: {dt * 1e3:.1f} ms ({rs.n / dt:.0f} reads/s) timings={ctx.timings()}", flush=True)
    assert np.array_equal(cls, cls2) and np.array_equal(strand, strand2)
if a.check:
    t = time.time()
    ocl, ost, ostat = oracle_entry_assignments(B, view)
    dt = time.time() - t
    print(f"oracle cluster {dt:.1f}s ({rs.n / dt:.0f} reads/s) {ostat}")
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    print("MISMATCHES:", len(bad), bad[:10])
    sys.exit(1 if len(bad) else 0)
