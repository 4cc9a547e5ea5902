#!/usr/bin/env python3
"""Developer timing of the GPU aligner: P pairs of LEN x LEN bases, cell updates per second."""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402

npairs, length = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(1)
base = bytes(rng.choice(b"ACGT") for _ in range(length))


def mutate(s, rate=0.1):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            out.append(rng.choice(b"ACGT"))
        elif x < 2 * rate / 3:
            continue
        elif x < rate:
            out += bytes([ch, rng.choice(b"ACGT")])
        else:
            out.append(ch)
    return bytes(out)


seqs = [mutate(base) for _ in range(min(npairs, 32) + 1)]
pairs = [(i % (len(seqs) - 1), i % (len(seqs) - 1) + 1, i % 2, 0.2) for i in range(npairs)]
ctx = api.Context(0)
ctx.align_set_pool(seqs)
cells = sum(len(seqs[a]) * len(seqs[b]) for a, b, _, _ in pairs)
for rep in range(3):
    t = time.time()
    score, win, ratio = ctx.align_pairs(pairs, 11)
    dt = time.time() - t
    print(f"{npairs} pairs of ~{length}: {dt * 1e3:.1f} ms, {cells / dt / 1e9:.1f} Gcells/s, ratio[0]={ratio[0]:.4f}", flush=True)
