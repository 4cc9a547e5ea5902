#!/usr/bin/env python3
"""Randomised differential run of consensus mode (toy graph store on both sides) against the oracle; the cases are
tests/fuzz_cases.py's.
    tools/fuzz_consensus.py [cases] [seed] [fast|sahlin]        (IOC_CONS_SPECULATE=0: every consensus at once)
    tools/fuzz_consensus.py --case "{'n': 162, ...}"           replay one case"""
import ast
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests import fuzz_cases as fz  # noqa: E402

ctx = api.Context(0)
if len(sys.argv) > 2 and sys.argv[1] == "--case":
    ok, why = fz.run_consensus(ctx, ast.literal_eval(sys.argv[2]))
    print("ok" if ok else f"MISMATCH {why}")
    sys.exit(0 if ok else 1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
mode = sys.argv[3] if len(sys.argv) > 3 else "fast"
bad, t0 = 0, time.time()
for case in range(n_cases):
    c = fz.draw_consensus(rng, mode)
    try:
        ok, why = fz.run_consensus(ctx, c)
    except Exception as e:  # noqa: BLE001
        ok, why = False, repr(e)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: --case \"{c}\"  ({why})", flush=True)
print(f"fuzz consensus: {n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
