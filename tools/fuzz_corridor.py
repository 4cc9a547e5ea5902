#!/usr/bin/env python3
"""Randomised soak of the corridor of the aligner's forward pass (tests/fuzz_cases.py::corridor_batch): long pairs of every kind,
the default build against IOC_ALIGN_CORRIDOR=0 and the host aligner.
    tools/fuzz_corridor.py [batches] [seed]      |      tools/fuzz_corridor.py --seed S"""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests import fuzz_cases as fz  # noqa: E402

ctx = api.Context(0)
if len(sys.argv) > 2 and sys.argv[1] == "--seed":
    ok, why = fz.corridor_batch(ctx, random.Random(int(sys.argv[2])))
    print("ok" if ok else f"MISMATCH {why}")
    sys.exit(0 if ok else 1)
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, t0 = 0, time.time()
for b in range(n_batches):
    ok, why = fz.corridor_batch(ctx, rng, npairs=24, host_checks=1)
    if not ok:
        bad += 1
        print("MISMATCH batch", b, why, flush=True)
    print(f"batch {b}: {bad} bad so far ({time.time() - t0:.0f} s)", flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
