#!/usr/bin/env python3
"""Developer timing of a merge at BASELINE config 4's batch shape: two sorted batches of N reads are clustered
on their own (fast mode), then the right one is merged into the left one (`cluster -l L -r R`)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, pipeline, synth  # noqa: E402
from tests.helpers import oracle_sorted_batch  # noqa: E402

n, g, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = api.Context(0)
p = api.default_params(11, 15, "fast")
cbs = []
for b in range(2):
    rs = synth.generate(n, g, L, 10, 21, seed=7)          # same transcripts in both batches
    if b:
        rs = synth.generate(n, g, L, 10, 21, seed=7 + 100 * b) if False else rs
    B, view = oracle_sorted_batch(rs)
    sb = pipeline.SortedBatch(view=view, read_ids=np.asarray(view["orig"]) + b * n, depth=-1, batch_start=b, batch_end=b)
    t = time.time()
    cb = pipeline.cluster_single(ctx, p, sb)
    print(f"batch {b}: {cb.n_clusters} clusters in {1e3 * (time.time() - t):.0f} ms (incl. export + bookkeeping)", flush=True)
    cbs.append(cb)
for r in range(2):
    t = time.time()
    m = pipeline.cluster_merge(ctx, p, cbs[0], cbs[1])
    print(f"merge: {m.n_clusters} clusters, {len(m.member_read)} members in {1e3 * (time.time() - t):.0f} ms; {m.stats}", flush=True)
