#!/usr/bin/env python3
"""Calibration of the corridor's width per couple: config 3's batch at several seeds, every aligned pair's
summed error rate e (ioc_aln_pair::e) against its score per base (IOC_V2_TRACE_RATES=1 on stderr -> gpurun_out/rates_seedN.txt).
    tools/corridor_rates.py FIRST_SEED N [config]"""
import os
import sys

sys.path.insert(0, ".")
os.environ["IOC_V2_TRACE_RATES"] = "1"
import bench  # noqa: E402
from isonclust2_amd import api, pipeline, synth  # noqa: E402

first, n = int(sys.argv[1]), int(sys.argv[2])
config = sys.argv[3] if len(sys.argv) > 3 else "config2"
for seed in range(first, first + n):
    ctx = api.Context(0)
    rs, sb, order = bench.prepare(ctx, api, pipeline, synth, config, seed, 11, 15, 0)
    sys.stderr.write(f"[seed] {seed}\n")
    sys.stderr.flush()
    cls, strand, st = ctx.cluster_resident()
    tm = ctx.timings()
    print(f"seed {seed}: clusters {len(set(cls.tolist()))}, forward {tm['ms_align_fwd']:.1f} ms, traceback {tm['ms_align_trace']:.1f} ms, "
          f"pairs {tm['n_align_pairs']}, cells computed {tm['n_align_cells_computed'] / max(1, tm['n_align_cells']):.3f} of the matrices", flush=True)
    ctx.close()
