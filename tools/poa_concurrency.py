#!/usr/bin/env python3
"""Developer timing: do the POA engine's latency-bound launches of SEVERAL clients overlap on one card?  N workers, each with a
context and an engine of its own, each adding ROUNDS rounds of one read per graph to G graphs (tools/poa_batch_bench.py's
workload) — as N threads of one process (streams of one process) or as N processes.
    tools/poa_concurrency.py threads|procs N [G] [LEN] [DEPTH] [ROUNDS]"""
import random
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")


def work(seed, G, length, depth, rounds, out, barrier=None):
    from isonclust2_amd import api
    from tests.test_gpu_poa import Poa, _mutate
    rng = random.Random(seed)
    truth = [bytes(rng.choice(b"ACGT") for _ in range(length)) for _ in range(G)]
    ctx = api.Context(0)
    poa = Poa(ctx)
    for g in range(G):
        poa.create(g, _mutate(rng, truth[g], 0.08))
    for d in range(depth):
        for g in range(G):
            poa.add(g, _mutate(rng, truth[g], 0.08))
        poa.graph(0)
    reads = [[_mutate(rng, truth[g], 0.08) for g in range(G)] for _ in range(rounds)]
    if barrier:
        barrier.wait()
    t = time.time()
    for r in range(rounds):
        for g in range(G):
            poa.add(g, reads[r][g])
        poa.graph(0)
    out.append(time.time() - t)
    poa.close()


if __name__ == "__main__":
    mode, N = sys.argv[1], int(sys.argv[2])
    G, length, depth, rounds = (int(x) for x in (sys.argv[3:7] + ["20", "2000", "8", "8"][len(sys.argv) - 3:]))
    if mode == "worker":
        out = []
        work(N, G, length, depth, rounds, out)
        print(f"{out[0]:.3f}")
    elif mode == "threads":
        out, bar = [], threading.Barrier(N)
        th = [threading.Thread(target=work, args=(s, G, length, depth, rounds, out, bar)) for s in range(N)]
        t = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        print(f"threads x{N}: rounds phase per worker {[round(v, 3) for v in out]} s (G {G}, {rounds} rounds); all {time.time() - t:.2f} s incl. setup")
    else:
        t = time.time()
        ps = [subprocess.Popen([sys.executable, __file__, "worker", str(s), str(G), str(length), str(depth), str(rounds)], stdout=subprocess.PIPE, text=True) for s in range(N)]
        out = [float(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        print(f"procs   x{N}: rounds phase per worker {[round(v, 3) for v in out]} s (G {G}, {rounds} rounds); all {time.time() - t:.2f} s incl. setup")
