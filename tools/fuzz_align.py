#!/usr/bin/env python3
"""Randomised soak of the GPU aligner against the host aligner (tests/fuzz_cases.py::align_batch): random lengths, related /
unrelated / low-complexity / identical sequences, every gap-open class, random k, random bands per pair.
    tools/fuzz_align.py [batches] [LMAX] [seed]      |      tools/fuzz_align.py --seed S   (one batch, as the test prints it)"""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests import fuzz_cases as fz  # noqa: E402

ctx = api.Context(0)
if len(sys.argv) > 2 and sys.argv[1] == "--seed":
    ok, why = fz.align_batch(ctx, random.Random(int(sys.argv[2])))
    print("ok" if ok else f"MISMATCH {why}")
    sys.exit(0 if ok else 1)
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2600
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
bad, t0 = 0, time.time()
for b in range(n_batches):
    ok, why = fz.align_batch(ctx, rng, lmax=lmax, npairs=40)
    if not ok:
        bad += 1
        print("MISMATCH batch", b, why, flush=True)
    print(f"... batch {b + 1}: {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz align: {n_batches * 40} pairs, {bad} bad batches")
sys.exit(1 if bad else 0)
