#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_assignments.json: FNV-1a digests of the oracle's cluster
assignments on the seeded synthetic configurations (regression fixtures for the oracle itself and
full-size references for the GPU path).  Run in the build container: python tools/gen_golden.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isonclust2_amd import synth  # noqa: E402
from tests.helpers import fnv1a, oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

CASES = [("tiny", 1), ("tiny", 7), ("config1", 1), ("config1", 3), ("short_dup", 1), ("short_dup", 2),
         ("short_dup", 3)]
if "--full" in sys.argv:
    CASES += [("config2", 1)]
out = {}
path = os.path.join("tests", "golden", "oracle_assignments.json")
if os.path.exists(path):
    out = json.load(open(path))
for name, seed in CASES:
    rs = synth.generate_config(name, seed=seed)
    B, view = oracle_sorted_batch(rs)
    cls, strand, st = oracle_entry_assignments(B, view)
    out[f"{name}:{seed}"] = {"n": rs.n, "bases": int(rs.offs[-1]), "clusters": B.n_clusters(),
                            "fnv1a": f"{fnv1a(cls, strand):016x}", "stats": st}
    print(name, seed, out[f"{name}:{seed}"])
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
