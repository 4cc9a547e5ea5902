#!/usr/bin/env python3
"""Randomised soak of the aligner's verdict mode (ioc_align_set_verdict_threshold: early-stopped tracebacks, walks parked
for the second launch with helper waves) against its own exact mode: random lengths up to LMAX (several 512-blocks, so that
walks do park), related / unrelated / low-complexity pairs, random thresholds, k and error classes.  Every comparison
`ratio >= threshold` and every score must come out the same; stopped walks never report more windows.
tools/fuzz_verdict.py [batches] [LMAX] [seed]"""
import random
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_align import _mutate  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
ctx = api.Context(0)
bad = 0
t0 = time.time()
for b in range(n_batches):
    seqs, pairs = [], []
    for t in range(120):
        n, m = rng.randint(0, lmax), rng.randint(0, lmax)
        if rng.random() < 0.5:
            m = max(0, n + rng.randint(-200, 200))
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 30))
        kind = rng.random()
        if kind < 0.45:
            q, r = _mutate(rng, base, rng.choice([0.02, 0.1, 0.25]))[:n], _mutate(rng, base[rng.randint(0, 20):], 0.1)[:m]
        elif kind < 0.8:
            q, r = bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(m))
        elif kind < 0.9:
            q, r = bytes(rng.choice(b"AC") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(m))
        else:   # related in one half only: decided late, either way
            q = base[:n]
            r = (base[: m // 2] + bytes(rng.choice(b"ACGT") for _ in range(m)))[:m]
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, rng.randint(0, 1), rng.choice([0.02, 0.05, 0.12, 0.3])))
    k = rng.choice([7, 11, 15])
    ctx.align_set_pool(seqs)
    ctx.align_set_verdict_threshold(0.0)
    s0, w0, r0 = ctx.align_pairs(pairs, k)
    for thr in (rng.choice([0.05, 0.2, 0.5]), rng.choice([0.1, 0.35, 0.9])):
        ctx.align_set_verdict_threshold(thr)
        s1, w1, r1 = ctx.align_pairs(pairs, k)
        ok = np.array_equal(s0, s1) and np.array_equal(r0 >= thr, r1 >= thr) and bool(np.all(w1 <= w0))
        if not ok:
            bad += 1
            x = np.nonzero((s0 != s1) | ((r0 >= thr) != (r1 >= thr)) | (w1 > w0))[0][:5]
            print("MISMATCH batch", b, "thr", thr, "k", k, [(int(i), len(seqs[2 * i]), len(seqs[2 * i + 1]), int(w0[i]), int(w1[i])) for i in x], flush=True)
    ctx.align_set_verdict_threshold(0.0)
    print(f"batch {b}: ok so far ({time.time() - t0:.0f} s)", flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
