#!/usr/bin/env python3
"""Randomised check of the POA engine's sequence-to-graph DP against the plain Python recurrence.

usage: fuzz_poa.py [graphs] [seed]   (each graph: 4-9 additions of mutated / cut / extended copies)"""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_poa import Poa, _mutate, _path_score, _ref_score  # noqa: E402

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = api.Context(0)
bad = adds = 0
t0 = time.time()
for g in range(n_graphs):
    poa = Poa(ctx)
    ln = rng.choice([40, 90, 200, 260, 330])
    truth = bytes(rng.choice(b"ACGT") for _ in range(ln))
    poa.create(0, _mutate(rng, truth, rng.choice([0.0, 0.05, 0.15])))
    for t in range(rng.randint(4, 9)):
        r = _mutate(rng, truth, rng.choice([0.02, 0.1, 0.25]))
        kind = rng.randint(0, 6)
        if kind == 0 and len(r) > 30:      # fragment
            a = rng.randint(0, len(r) // 2)
            r = r[a:a + rng.randint(10, len(r) - a)]
        elif kind == 1:                    # long deletion (edges spanning many rows)
            a = rng.randint(0, max(1, len(r) - 60))
            r = r[:a] + r[a + rng.randint(17, 60):]
        elif kind == 2:                    # long insertion
            a = rng.randint(0, len(r))
            r = r[:a] + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(17, 70))) + r[a:]
        elif kind == 3:                    # unrelated head / tail
            r = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(5, 40))) + r + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 40)))
        if not r:
            continue
        bases, rank, ef, et, ew = poa.graph(0)
        want = _ref_score(bases, rank, ef, et, r)
        poa.add(0, r, w=1 + t % 3)
        nodes, pos, score = poa.last_alignment()
        adds += 1
        ok = score == want and _path_score(bases, ef, et, r, nodes, pos) == score
        if not ok:
            bad += 1
            print(f"MISMATCH graph {g} add {t}: device {score}, recurrence {want}, nodes {len(bases)}, read {len(r)}", flush=True)
    poa.close()
    print(f"graph {g}: {adds} additions checked, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {adds} additions, {bad} mismatches")
sys.exit(1 if bad else 0)
