#!/usr/bin/env python3
"""Randomised differential run against the oracle (developer soak): random batch shapes, qualities, duplicated
transcripts, (k, w), fast mode; single batches and two-batch merges.  Prints every mismatch and a summary."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, pipeline, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
aln_mode = sys.argv[3] if len(sys.argv) > 3 else None     # "sahlin" / "furious": small batches through the alignment fallback
rng = np.random.default_rng(seed0)
ctx = api.Context(0)
# (sahlin / furious: the oracle aligns with its own scalar aligner, oracle.cpp sg_trace — nothing of the product behind it)
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.integers(1, 260))
    g = int(rng.integers(1, 24))
    ln = int(rng.choice([120, 200, 350, 600, 900, 1500, 2500]))
    qlo = float(rng.choice([7, 9, 11, 14]))
    qhi = qlo + float(rng.choice([4, 8, 12]))
    dup = int(rng.choice([0, 0, 2, 3]))
    jit = float(rng.choice([0.0, 0.0, 0.3]))
    k, w = [(11, 15), (11, 15), (13, 20), (10, 14), (15, 22)][int(rng.integers(0, 5))]
    seed = int(rng.integers(0, 1 << 30))
    if aln_mode:
        n, ln = min(n, 70), min(ln, 350)
    rs = synth.generate(n, g, ln, qlo, qhi, seed=seed, dup_every=dup, len_jitter=jit)
    tag = f"case {case}: n={n} g={g} L={ln} Q=[{qlo},{qhi}] dup={dup} jit={jit} k={k} w={w} seed={seed}"
    try:
        B, view = oracle_sorted_batch(rs, k, w)
        if aln_mode:
            ocl, ost, _ = oracle_entry_assignments(B, view, mode=aln_mode)
            seqs = [rs.read(int(i))[0] for i in view["orig"]]
            off = np.zeros(len(seqs) + 1, np.int64)
            off[1:] = np.cumsum([len(x) for x in seqs])
            v2 = dict(view)
            v2.update(raw_seq=b"".join(seqs), raw_off=off)
            cls, strand, st = ctx.cluster_batch(api.default_params(k, w, aln_mode), v2)
        else:
            ocl, ost, _ = oracle_entry_assignments(B, view)
            cls, strand, st = ctx.cluster_batch(api.default_params(k, w, "fast"), view)
        if not (np.array_equal(cls, ocl) and np.array_equal(strand, ost)):
            bad += 1
            d = np.nonzero((cls != ocl) | (strand != ost))[0]
            print("MISMATCH", tag, "first entries", d[:5], cls[d[:5]], ocl[d[:5]], flush=True)
        if case % 4 == 0 and n >= 8 and not aln_mode:   # a two-batch merge on the same reads
            R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
            R.score_sort(k, w)
            p = po.default_params(k, w)
            cut = n // 2
            obs, cbs = [], []
            for b, (lo, hi) in enumerate(((0, cut - 1), (cut, n - 1))):
                Bo = po.Batch(R, lo, hi, p, batch_nr=b)
                info, off_f, off_r, mn, ps = Bo.minimizer_soa()
                vw = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"], hpc_len=info["hpc_len"],
                          score=info["score"], raw_err=info["raw_err"], hpc_err=info["hpc_err"],
                          state=info["state"].astype(np.uint8), min_qual=p.min_qual)
                sb = pipeline.SortedBatch(view=vw, read_ids=info["orig"].astype(np.int64), batch_nr=b, batch_start=lo, batch_end=hi)
                Bo.cluster(mode="fast")
                obs.append(Bo)
                cbs.append(pipeline.cluster_single(ctx, api.default_params(k, w, "fast"), sb))
            obs[0].cluster(right=obs[1], mode="fast")
            merged = pipeline.cluster_merge(ctx, api.default_params(k, w, "fast"), cbs[0], cbs[1])
            mo, ms = obs[0].assignments(rs.n)
            mc, mst = merged.assignments(rs.n)
            if not (np.array_equal(mc, mo) and np.array_equal(mst, ms)):
                bad += 1
                d = np.nonzero((mc != mo) | (mst != ms))[0]
                print("MERGE MISMATCH", tag, "first reads", d[:5], mc[d[:5]], mo[d[:5]], flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(e), flush=True)
    if case % 50 == 49:
        print(f"... {case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
