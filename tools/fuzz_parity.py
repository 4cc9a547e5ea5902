#!/usr/bin/env python3
"""Randomised differential run against the oracle (developer soak): random batch shapes, qualities, duplicated
transcripts, (k, w); single batches and (fast mode) two-batch merges.  The cases are tests/fuzz_cases.py's — the same ones
tests/test_gpu_fuzz.py runs in bounded slices.
    tools/fuzz_parity.py [cases] [seed] [sahlin|furious]
    tools/fuzz_parity.py --case "{'n': 162, ...}"      replay one case (the reproducer a failing test prints)"""
import ast
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests import fuzz_cases as fz  # noqa: E402

ctx = api.Context(0)
if len(sys.argv) > 2 and sys.argv[1] == "--case":
    c = ast.literal_eval(sys.argv[2])
    ok, why = fz.run_parity(ctx, c, merge=True)
    print("ok" if ok else f"MISMATCH {why}")
    sys.exit(0 if ok else 1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
aln_mode = sys.argv[3] if len(sys.argv) > 3 else None
bad, t0 = 0, time.time()
for case in range(n_cases):
    c = fz.draw_parity(rng, aln_mode)
    try:
        ok, why = fz.run_parity(ctx, c, merge=(case % 4 == 0))
    except Exception as e:   # noqa: BLE001
        ok, why = False, repr(e)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: --case \"{c}\"  ({why})", flush=True)
    if case % 50 == 49:
        print(f"... {case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
