#!/usr/bin/env python3
"""Developer aid: N pairs of related 16.7 kb sequences through the aligner with and without the row-by-row tile dependencies of
launches with few couples (IOC_ALIGN_V2_FEW): where does waiting for rows stop paying?   tools/few_limit.py N [N ...]"""
import os
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_align import _mutate  # noqa: E402

rng = random.Random(7)
base = [bytes(rng.choice(b"ACGT") for _ in range(16700)) for _ in range(8)]
ctx = api.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [64, 128, 256, 512]:
    seqs = []
    for t in range(n):
        seqs += [_mutate(rng, base[t % 8], 0.06), _mutate(rng, base[t % 8], 0.07)]
    pairs = [(2 * t, 2 * t + 1, 0, 0.12) for t in range(n)]
    ctx.align_set_pool(seqs)
    ctx.align_set_verdict_threshold(0.2)
    for few in ("0", "100000"):
        os.environ["IOC_ALIGN_V2_FEW"] = few
        ctx.align_pairs(pairs, 11)
        t0 = time.time()
        ctx.align_pairs(pairs, 11)
        print(f"{n} pairs, IOC_ALIGN_V2_FEW={few}: {1e3 * (time.time() - t0):.1f} ms", flush=True)
    ctx.align_set_verdict_threshold(0.0)
