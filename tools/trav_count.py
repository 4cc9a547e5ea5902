#!/usr/bin/env python3
"""Developer count: posting slots the scoring kernel walks for the config-2 batch (IOC_COUNT_TRAVERSED), next to the
postings the reference traverses.  With a -DIOC_SCORE_TRAV_CAPACITY=1 build (IOC_LIB=...) the count is the slots of the
wave steps, filled or not."""
import os
import sys

os.environ["IOC_COUNT_TRAVERSED"] = "1"
sys.path.insert(0, ".")
from isonclust2_amd import api, pipeline, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
rs = synth.generate_config("config2", seed=1)
ctx = api.Context(0)
sb, _ = pipeline.sort_stage(ctx, rs, 11, 15)
cb = pipeline.cluster_single(ctx, api.default_params(11, 15, mode), sb)
t = ctx.timings()
print("postings_traversed (slots, wave-summed lanes count once per list unit):", t["postings_traversed"], "clusters", cb.n_clusters)
