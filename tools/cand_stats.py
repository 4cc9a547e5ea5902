import sys, numpy as np
sys.path.insert(0, ".")
import torch
from isonclust2_amd import api, pipeline, synth
import bench
ctx = api.Context(0)
rs, sb, order = bench.prepare(ctx, api, pipeline, synth, "config2", 1, 11, 15, 0)
for mode in ("fast", "sahlin"):
    ctx.set_params(api.default_params(11, 15, mode))
    cls, strand, st = ctx.cluster_resident()
    n = rs.n
    lens = []
    for q in range(0, n, 7):
        k, z = ctx.scored_candidates(q, 2 * n + 2)
        lens.append(len(k))
    lens = np.array(lens)
    print(mode, "queries sampled", len(lens), "mean candidates", lens.mean(), "median", np.median(lens), "max", lens.max(), "est total", lens.mean() * n, ctx.timings()["n_mapped_evals"])
