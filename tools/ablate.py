#!/usr/bin/env python3
"""Developer tool: times ioc_score under the ablation variants of k_score (IOC_SCORE_VARIANT)."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402
from bench import prepare_resident_batch  # noqa: E402

ctx = api.Context(0)
rs, order, nmin = prepare_resident_batch(ctx, api, synth, "config2", 1, 11, 15)
ctx.index_build()
for name, env in [("full", {}), ("no-atomics", {"IOC_SCORE_VARIANT": "1"}), ("no-posting-loads", {"IOC_SCORE_VARIANT": "2"}),
                  ("probes-only", {"IOC_SCORE_VARIANT": "4"}), ("no-loads-no-atomics", {"IOC_SCORE_VARIANT": "6"}), ("no-loads-stores-not-atomics", {"IOC_SCORE_VARIANT": "7"}), ("no-probes-no-traversal", {"IOC_SCORE_VARIANT": "5"}), ("partitioned", {"IOC_SCORE_PARTS": "1"}), ("full", {})]:
    for k in ("IOC_SCORE_VARIANT", "IOC_SCORE_PARTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for _ in range(4):
        ctx.score()
        ts.append(ctx.timings()["ms_score"])
    print(f"{name:18s} ms_score {min(ts):.3f} (median {np.median(ts):.3f})", flush=True)
