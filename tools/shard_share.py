"""One rank's share of a sharded fast-mode merge, timed on ONE card (profiles/r03_shard_share.json).

The merge of 8 freshly clustered config2 batches (the representatives of 8 ranks: BASELINE.json configs[3]) runs first
unsharded, then with ioc_set_shard(world, 0, noop) for world = 2, 4, 8: rank 0's share of the scoring is exactly what it
would run next to 7 peers (scoring does not read `valid`); the resolve's sweeps see a `valid` without the peers' shares, so
their number and length are indicative only, and the call may end in an error after the resolve (no peer answers the tie
replays): the device timings are read either way.  What a real exchange adds is two all-reduces per sweep plus four at the
end (ioc_shard_exchanges), each a small in-stream RCCL all-reduce."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isonclust2_amd import api, dist as d, pipeline, synth  # noqa: E402


def main():
    ctx = api.Context(0)
    p = api.default_params(11, 15, "fast")
    batches = []
    for seed in range(1, 9):
        rs = synth.generate_config("config2", seed=seed)
        sb, _ = pipeline.sort_stage(ctx, rs, 11, 15, read_id_base=(seed - 1) * rs.n, batch_nr=seed - 1)
        sb.batch_start, sb.batch_end = (seed - 1) * rs.n, seed * rs.n - 1
        batches.append(pipeline.cluster_single(ctx, p, sb))
    out = {"representatives": int(sum(b.n_clusters for b in batches)), "legs": []}
    for world in (1, 2, 4, 8):
        ctx.set_shard(world, 0, (lambda *a: 0) if world > 1 else None)
        best = None
        for _ in range(3):
            err = None
            try:
                m = d.merge_all(ctx, p, batches, export_mindb=False)
            except Exception as e:  # noqa: BLE001
                err = str(e)[:120]
            t = ctx.timings()
            rec = dict(world=world, ms_build=t["ms_build"], ms_score=t["ms_score"], ms_resolve=t["ms_resolve"], sweeps=t["resolve_iters"],
                       exchanges=ctx.shard_exchanges, clusters_out=None if err else m.n_clusters, error=err)
            if best is None or rec["ms_score"] + rec["ms_resolve"] < best["ms_score"] + best["ms_resolve"]:
                best = rec
        out["legs"].append(best)
        print(json.dumps(best), flush=True)
    ctx.set_shard(1, 0, None)
    json.dump(out, open(os.path.join("gpurun_out", "shard_share.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
