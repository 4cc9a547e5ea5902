#!/usr/bin/env python3
"""Developer aid: the aligner with and without the corridor on a few long pairs (scores, windows)."""
import os
import random
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_align import _mutate  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
rng = random.Random(5)
base = bytes(rng.choice(b"ACGT") for _ in range(L))
other = bytes(rng.choice(b"ACGT") for _ in range(L))
seqs = [_mutate(rng, base, rate) for _ in range(4)] + [other]
pairs = [(0, 1, 0, 0.12), (2, 3, 0, 0.12), (1, 2, 0, 0.12), (0, 4, 0, 0.12), (3, 0, 0, 0.12)]
ctx = api.Context(0)
ctx.align_set_pool(seqs)
out = {}
for frac in ("0", "0.2"):
    os.environ["IOC_ALIGN_CORRIDOR"] = frac
    out[frac] = ctx.align_pairs(pairs, 11)
    print("corridor", frac, "scores", out[frac][0].tolist(), "windows", out[frac][1].tolist(), flush=True)
print("equal:", all(np.array_equal(a, b) for a, b in zip(out["0"], out["0.2"])))
