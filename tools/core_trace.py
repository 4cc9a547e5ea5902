import sys, time
sys.path.insert(0, ".")
import numpy as np
from isonclust2_amd import api, pipeline, synth
ctx = api.Context(0)
rs = synth.generate_config("config2", seed=1)
sb, order = pipeline.sort_stage(ctx, rs, 11, 15)
p = api.default_params(11, 15, sys.argv[1] if len(sys.argv) > 1 else "sahlin")
for i in range(3):
    tm = {}
    t0 = time.perf_counter()
    cb = pipeline.cluster_single(ctx, p, sb, timing=tm)
    print("run", i, tm, flush=True)
