#!/usr/bin/env python3
"""Randomised soak of the GPU aligner against the host aligner: random lengths (0..LMAX), related / unrelated /
low-complexity / identical sequences, every gap-open class, random k, random bands per pair."""
import os
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import _lib, api  # noqa: E402
from tests.test_gpu_align import _host, _mutate  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2600
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
L = _lib.load()
ctx = api.Context(0)
bad = 0
t0 = time.time()
for b in range(n_batches):
    seqs, pairs = [], []
    for t in range(40):
        n, m = rng.choice([0, 1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257]) if rng.random() < 0.3 else rng.randint(0, lmax), rng.randint(0, lmax)
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 30))
        kind = rng.random()
        if kind < 0.5:
            q, r = _mutate(rng, base, rng.choice([0.02, 0.1, 0.25]))[:n], _mutate(rng, base[rng.randint(0, 20):], 0.1)[:m]
        elif kind < 0.7:
            q, r = bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(m))
        elif kind < 0.85:
            q, r = bytes(rng.choice(b"AC") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(m))
        else:
            q = base[:n]
            r = q[: m] if rng.random() < 0.5 else (b"ACGT" * (m // 4 + 1))[:m]
        if rng.random() < 0.05:
            q = q[: len(q) // 2] + b"N" + q[len(q) // 2 + 1:]      # a letter outside A C G T: the comparing kernel
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, rng.randint(0, 1), rng.choice([0.0, 0.02, 0.05, 0.12, 0.3, 0.95])))
    k = rng.choice([1, 7, 11, 15, 32])
    os.environ["IOC_ALIGN_WAVES"] = rng.choice(["1", "2", "4", "8"])
    ctx.align_set_pool(seqs)
    score, win, ratio = ctx.align_pairs(pairs, k)
    for i, (qi, ri, rc, e) in enumerate(pairs):
        hs, hr = _host(L, seqs[qi], seqs[ri], rc, e, k)
        if score[i] != hs or ratio[i] != hr:
            bad += 1
            print("MISMATCH batch", b, "pair", i, len(seqs[qi]), len(seqs[ri]), rc, e, k, os.environ["IOC_ALIGN_WAVES"], score[i], hs, ratio[i], hr, flush=True)
    print(f"... batch {b + 1}: {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz align: {n_batches * 40} pairs, {bad} bad")
