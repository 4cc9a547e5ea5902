#!/usr/bin/env python3
"""Developer timing of configs[3] on ONE GPU: the 8 batches of seeds 1..8 clustered one after the other, then merged —
step by step (7 merges) and as one pass (dist.merge_all) — with the digests compared against the golden."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from isonclust2_amd import api, dist, pipeline, synth  # noqa: E402
from isonclust2_amd.digest import fnv1a_reads  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
gold = json.load(open(os.path.join("tests", "golden", "oracle_assignments.json")))
ctx = api.Context(0)
p = api.default_params(11, 15, mode)
import torch  # noqa: E402
dev = torch.device("cuda", 0)
cbs, parts, metas = [], [], []
for seed in range(1, nb + 1):
    rs = synth.generate_config("config2", seed=seed)
    sb, _ = pipeline.sort_stage(ctx, rs, 11, 15, read_id_base=rs.n * (seed - 1), batch_nr=seed - 1)
    t0 = time.perf_counter()
    cbs.append(pipeline.cluster_single(ctx, p, sb))
    t1 = time.perf_counter()
    parts.append(dist.gather_local(ctx, cbs[-1], torch, dev))
    metas.append(dist.unpack_clustered(dist.pack_clustered(cbs[-1], with_minimizers=False)))
    print(f"seed {seed}: {cbs[-1].n_clusters} clusters, {1e3 * (t1 - t0):.1f} ms (host arrays -> ClusteredBatch), "
          f"{1e3 * (time.perf_counter() - t1):.1f} ms device gather + meta pack", flush=True)
t0 = time.perf_counter()
a = dist.fold_merge(ctx, p, cbs)
t1 = time.perf_counter()
b = dist.merge_all(ctx, p, cbs)
t2 = time.perf_counter()
cap = max(int(m.numel()) for m, _ in parts)
pad = lambda t: torch.cat([t, torch.zeros(cap - int(t.numel()), dtype=torch.int32, device=dev)])
recv_min = torch.cat([pad(m) for m, _ in parts])
recv_pos = torch.cat([pad(q) for _, q in parts])
torch.cuda.synchronize()
td0 = time.perf_counter()
c = dist.merge_gathered(ctx, p, metas, recv_min, recv_pos, cap)
td1 = time.perf_counter()
tm = {}
t3 = time.perf_counter()
pk = [dist.pack_clustered(c, with_mindb=(i == 0)) for i, c in enumerate(cbs)]
t4 = time.perf_counter()
un = [dist.unpack_clustered(x) for x in pk]
t5 = time.perf_counter()
key = f"config4:{mode}"
want = gold.get(key, {}).get("fnv1a") if nb == 8 else None
print(json.dumps({"mode": mode, "batches": nb, "fold_ms": (t1 - t0) * 1e3, "one_pass_ms": (t2 - t1) * 1e3, "clusters": [a.n_clusters, b.n_clusters],
                  "device_records_ms": (td1 - td0) * 1e3,
                  "digests": [fnv1a_reads(a), fnv1a_reads(b), fnv1a_reads(c), want], "pack_ms": (t4 - t3) * 1e3, "unpack_ms": (t5 - t4) * 1e3,
                  "payload_MB": [round(4 * len(x) / 1e6, 1) for x in pk], "timings_last": ctx.timings()}))
