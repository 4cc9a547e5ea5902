#!/usr/bin/env python3
"""Developer timing of the POA engine: N noisy copies of one LEN-base sequence into one graph."""
import random
import sys
import time

sys.path.insert(0, ".")
from isonclust2_amd import api  # noqa: E402
from tests.test_gpu_poa import Poa, _mutate  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(1)
truth = bytes(rng.choice(b"ACGT") for _ in range(length))
ctx = api.Context(0)
poa = Poa(ctx)
poa.create(0, _mutate(rng, truth, 0.1))
for i in range(n):
    r = _mutate(rng, truth, 0.1)
    t = time.time()
    poa.add(0, r)
    nodes = len(poa.graph(0)[0])  # additions are queued: reading the graph works the queue off
    dt = time.time() - t
    print(f"add {i}: {dt * 1e3:.1f} ms, graph {nodes} nodes, {nodes * len(r) / dt / 1e9:.2f} Gcells/s", flush=True)
t = time.time()
c = poa.consensus(0)
print(f"consensus {len(c)} bases in {(time.time() - t) * 1e3:.1f} ms")
