#!/usr/bin/env python3
"""Developer timing of the POA engine on the shape of a consensus leaf's rounds: G graphs of noisy copies of G different LEN-base
sequences, DEPTH additions each beforehand, then ROUNDS rounds of one addition per graph worked off as ONE batch.
IOC_TRACE=1 prints the engine's own counters at the end.   tools/poa_batch_bench.py [G] [LEN] [DEPTH] [ROUNDS]"""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_poa import Poa, _mutate  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100
length = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = random.Random(1)
truth = [bytes(rng.choice(b"ACGT") for _ in range(length)) for _ in range(G)]
ctx = api.Context(0)
poa = Poa(ctx)
for g in range(G):
    poa.create(g, _mutate(rng, truth[g], 0.08))
t = time.time()
for d in range(depth):
    for g in range(G):
        poa.add(g, _mutate(rng, truth[g], 0.08))
    poa.graph(0)
print(f"{G} graphs x {depth} additions of {length} bases: {time.time() - t:.2f} s; graph 0 has {len(poa.graph(0)[0])} nodes", flush=True)
for r in range(rounds):
    for g in range(G):
        poa.add(g, _mutate(rng, truth[g], 0.08))
    t = time.time()
    nodes = len(poa.graph(0)[0])
    print(f"round {r}: {1e3 * (time.time() - t):.2f} ms for {G} additions (graph 0: {nodes} nodes)", flush=True)
poa.close()
