#!/usr/bin/env python3
"""Developer timing of a small alignment batch (the later rounds of a sahlin batch): N full-length pairs."""
import random
import sys

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = synth.generate_config("config2", seed=1)
rng = random.Random(5)
ids = rng.sample(range(rs.n), 2 * n)
ctx = api.Context(0)
ctx.align_set_pool([bytes(rs.read(i)[0]) for i in ids])
pairs = [(2 * t, 2 * t + 1, t % 2, 0.2) for t in range(n)]
ctx.align_pairs(pairs, 11)
for r in range(3):
    t0 = ctx.timings()
    s, w, _ = ctx.align_pairs(pairs, 11)
    t1 = ctx.timings()
    print(f"{n} pairs: fwd {t1['ms_align_fwd'] - t0['ms_align_fwd']:.2f} ms, trace {t1['ms_align_trace'] - t0['ms_align_trace']:.2f} ms, checksum {int(w.sum())}", flush=True)
