#!/usr/bin/env python3
"""Developer aid: config 3's batch at several seeds through one resident sahlin step each (IOC_TRACE=1 on stderr says how many
pairs the corridor's certificate refuted, i.e. ran again): does a corridor fraction hold beyond the bench's seed?
    tools/corridor_seeds.py FIRST_SEED N"""
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from isonclust2_amd import api, pipeline, synth  # noqa: E402

first, n = int(sys.argv[1]), int(sys.argv[2])
config = sys.argv[3] if len(sys.argv) > 3 else "config2"
for seed in range(first, first + n):
    ctx = api.Context(0)
    rs, sb, order = bench.prepare(ctx, api, pipeline, synth, config, seed, 11, 15, 0)
    cls, strand, st = ctx.cluster_resident()
    cls, strand, st = ctx.cluster_resident()
    tm = ctx.timings()
    print(f"seed {seed}: clusters {len(set(cls.tolist()))}, forward {tm['ms_align_fwd']:.1f} ms, traceback {tm['ms_align_trace']:.1f} ms, "
          f"pairs {tm['n_align_pairs']}, cells computed {tm['n_align_cells_computed'] / max(1, tm['n_align_cells']):.3f} of the matrices", flush=True)
    ctx.close()
