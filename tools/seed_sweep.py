#!/usr/bin/env python3
"""Developer check of the N > 1 bench workloads: the config-2 batch of rank r uses seed 1 + r.  For a few
seeds: fast mode against the oracle on the full batch, sahlin mode against the oracle on a sample, timings."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from isonclust2_amd import api, synth  # noqa: E402

seeds = [int(x) for x in sys.argv[1:]] or [2, 5, 8]
for seed in seeds:
    ctx = api.Context(0)
    rs, order, n_min = bench.prepare_resident_batch(ctx, api, synth, "config2", seed, 11, 15, "sahlin")
    ctx.set_params(api.default_params(11, 15, "fast"))
    cls, strand, st = ctx.cluster_resident()
    dt, ost, mism = bench.cpu_baseline_and_parity(rs, order, cls, strand, 11, 15)
    ctx.set_params(api.default_params(11, 15, "sahlin"))
    ctx.cluster_resident()
    t = time.perf_counter()
    scls, sstrand, sst = ctx.cluster_resident()
    ms = (time.perf_counter() - t) * 1e3
    res = bench.cpu_baseline_sahlin_sample(rs, order, scls, sstrand, 11, 15, 20)
    print(f"seed {seed}: fast clusters {st['n_clusters']} mismatches {mism} (oracle {dt:.1f} s); sahlin {ms:.1f} ms, "
          f"clusters {sst['n_clusters']}, pairs {sst['n_aln_pairs']}, rounds {sst['aln_rounds']}, "
          f"sample mismatches {res[2]} of {res[3]}", flush=True)
    ctx.close()
