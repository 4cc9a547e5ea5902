#!/usr/bin/env python3
"""Developer check at larger N: many short reads (range passes, big candidate tables), HIP vs oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

n, g, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = synth.generate(n, g, L, 10, 21, seed=3)
t = time.time()
B, view = oracle_sorted_batch(rs)
print(f"prep {time.time() - t:.1f}s, minimizers {len(view['min_val'])}", flush=True)
ctx = api.Context(0)
p = api.default_params(11, 15, "fast")
for r in range(2):
    t = time.time()
    cls, strand, st = ctx.cluster_batch(p, view)
    print(f"hip {1e3 * (time.time() - t):.1f} ms {st} {ctx.timings()}", flush=True)
t = time.time()
ocl, ost, ostat = oracle_entry_assignments(B, view)
print(f"oracle {time.time() - t:.1f}s {ostat}")
bad = np.nonzero((cls != ocl) | (strand != ost))[0]
print("MISMATCHES", len(bad), bad[:10])
sys.exit(1 if len(bad) else 0)
