#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_assignments.json: FNV-1a digests of the oracle's cluster assignments on the
seeded synthetic configurations (regression fixtures for the oracle itself and full-size references for the
GPU path).  Run in the build container:

    python tools/gen_golden.py                 small cases (seconds)
    python tools/gen_golden.py --full          + BASELINE.json configs[1] (config2:1, fast) and configs[3] in fast mode
                                                 (the 8 batches of seeds 1..8 clustered one by one, then folded left
                                                 to right like the reference README's example, README.md:105-117)
    python tools/gen_golden.py --full-sahlin [--threads T]
                                               + configs[2] (config2:1, sahlin) and configs[3] in sahlin mode; the
                                                 alignment fallback is the oracle's own scalar aligner (~0.5 s per
                                                 16.7 kb pair, ~1600 pairs per batch): T batches at a time, ~15 min each

Keys: "<config>:<seed>" (fast), "<config>:<seed>:sahlin", "config4:fast", "config4:sahlin".
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from isonclust2_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import fnv1a, oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "oracle_assignments.json")
SEEDS = list(range(1, 9))
out = json.load(open(PATH)) if os.path.exists(PATH) else {}
lock = threading.Lock()


def save():
    with lock:
        json.dump(out, open(PATH, "w"), indent=1, sort_keys=True)


def small():
    for name, seed in [("tiny", 1), ("tiny", 7), ("config1", 1), ("config1", 3), ("short_dup", 1), ("short_dup", 2),
                       ("short_dup", 3)]:
        rs = synth.generate_config(name, seed=seed)
        B, view = oracle_sorted_batch(rs)
        cls, strand, st = oracle_entry_assignments(B, view)
        out[f"{name}:{seed}"] = {"n": rs.n, "bases": int(rs.offs[-1]), "clusters": B.n_clusters(),
                                "fnv1a": f"{fnv1a(cls, strand):016x}", "stats": st}
        print(name, seed, out[f"{name}:{seed}"], flush=True)
    for name, seed in [("tiny", 7), ("config1", 1), ("short_dup", 1)]:
        rs = synth.generate_config(name, seed=seed)
        B, view = oracle_sorted_batch(rs)
        cls, strand, st = oracle_entry_assignments(B, view, mode="sahlin")
        out[f"{name}:{seed}:sahlin"] = {"n": rs.n, "bases": int(rs.offs[-1]), "clusters": B.n_clusters(),
                                       "fnv1a": f"{fnv1a(cls, strand):016x}", "stats": st}
        print(name, seed, "sahlin", out[f"{name}:{seed}:sahlin"], flush=True)
    save()


def seed_batch(seed, k=11, w=15):
    """The batch of config2 / seed, sorted on its own (one batch per GPU in configs[3]); read ids 3000*(seed-1)+."""
    rs = synth.generate_config("config2", seed=seed)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    order, _, _ = R.order()
    R.shift_orig(rs.n * (seed - 1))
    B = po.Batch(R, 0, rs.n - 1, po.default_params(k, w), batch_nr=seed - 1)
    return rs, R, B, order.astype(np.int64)


def one_seed(seed, mode, store):
    rs, R, B, order = seed_batch(seed)
    t0 = time.perf_counter()
    st = B.cluster(mode=mode)
    dt = time.perf_counter() - t0
    base = rs.n * (seed - 1)
    acl, ast = B.assignments(base + rs.n)
    cls, strand = acl[base + order], ast[base + order]          # per entry of the sorted batch
    key = f"config2:{seed}" + ("" if mode == "fast" else f":{mode}")
    with lock:
        out[key] = {"n": rs.n, "bases": int(rs.offs[-1]), "clusters": B.n_clusters(),
                    "fnv1a": f"{fnv1a(cls, strand):016x}", "stats": st, "oracle_seconds_1core": round(dt, 2)}
        print(key, out[key], flush=True)
    save()
    store[seed] = (R, B, rs.n)


def fold(store, mode):
    """((b1 + b2) + b3) ... : `cluster -l L -r R` left to right (README.md:105-117; src/cluster.cpp:67-322)."""
    left = store[SEEDS[0]][1]
    steps = []
    total = sum(store[s][2] for s in SEEDS)
    for s in SEEDS[1:]:
        t0 = time.perf_counter()
        st = left.cluster(right=store[s][1], mode=mode)
        steps.append({"right_seed": s, "clusters_after": left.n_clusters(), "stats": st,
                      "oracle_seconds_1core": round(time.perf_counter() - t0, 2)})
        print("fold", mode, steps[-1], flush=True)
    acl, ast = left.assignments(total)
    out[f"config4:{mode}"] = {"seeds": SEEDS, "n": total, "clusters": left.n_clusters(),
                              "assigned": int(np.count_nonzero(acl >= 0)),
                              "fnv1a": f"{fnv1a(acl, ast):016x}", "steps": steps,
                              "note": "digest over (cluster, strand) of reads 0..n-1, read id = 3000*(seed-1) + index in the seed's generator order"}
    print(f"config4:{mode}", {k: v for k, v in out[f'config4:{mode}'].items() if k != "steps"}, flush=True)
    save()


def run_mode(mode, threads):
    store = {}
    pending = list(SEEDS)
    def worker():
        while True:
            with lock:
                if not pending:
                    return
                s = pending.pop(0)
            one_seed(s, mode, store)
    ts = [threading.Thread(target=worker) for _ in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    fold(store, mode)


if __name__ == "__main__":
    threads = int(sys.argv[sys.argv.index("--threads") + 1]) if "--threads" in sys.argv else 4
    if "--full" not in sys.argv and "--full-sahlin" not in sys.argv:
        small()
    if "--full" in sys.argv:
        run_mode("fast", 1)       # one at a time: the per-batch seconds double as the 1-core CPU calibration
    if "--full-sahlin" in sys.argv:
        run_mode("sahlin", threads)
