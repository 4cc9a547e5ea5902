#!/usr/bin/env python3
"""Randomised soak of the consensus driver's deferred mode (ioc_consensus_spec_ops, POA engine as the graph store)
against the same driver taking every consensus at once: random batch shapes, modes, consensus parameters, window
sizes, and forced rollbacks.  Compared: assignments, every representative replacement with its consensus sequence,
the final MinDB, every final graph."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402
from tests.test_gpu_poa import _run_consensus  # noqa: E402


class Env:   # the monkeypatch interface _run_consensus uses
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=True):
        os.environ.pop(k, None)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = api.Context(0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.integers(20, 420))
    g = int(rng.integers(1, 14))
    ln = int(rng.choice([300, 500, 800, 1200]))
    mode = str(rng.choice(["fast", "fast", "sahlin"]))
    cons = (int(rng.choice([2, 3, 5, 20])), int(rng.choice([3, 6, 12, 50, 1000])), int(rng.choice([5, 25, 500])))
    window = [None, None, 5, 16, 64][int(rng.integers(0, 5))]
    force = ["", "", "2", "3", "7"][int(rng.integers(0, 5))]
    seed = int(rng.integers(0, 1 << 30))
    if mode == "sahlin":
        n, ln = min(n, 160), min(ln, 800)
    rs = synth.generate(n, g, ln, 11, 22, seed=seed, dup_every=int(rng.choice([0, 0, 2])))
    tag = f"case {case}: n={n} g={g} L={ln} {mode} cons={cons} window={window} force={force or '-'} seed={seed}"
    try:
        a = _run_consensus(ctx, rs, mode, cons, window, speculate=False, monkeypatch=Env())
        if force:
            os.environ["IOC_CONS_FORCE_ROLLBACK"] = force
        b = _run_consensus(ctx, rs, mode, cons, window, speculate=True, monkeypatch=Env())
        os.environ.pop("IOC_CONS_FORCE_ROLLBACK", None)
        ok = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3] and all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
              and a[5] == b[5] and a[2]["n_cons_invoked"] == b[2]["n_cons_invoked"])
    except Exception as e:  # noqa: BLE001
        os.environ.pop("IOC_CONS_FORCE_ROLLBACK", None)
        ok = False
        print(tag, "EXCEPTION", repr(e)[:300], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", tag, flush=True)
    if case % 10 == 9:
        print(f"... {case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz consensus (deferred vs immediate, POA engine): {n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
