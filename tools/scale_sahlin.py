#!/usr/bin/env python3
"""Developer timing of sahlin mode on a batch of many short reads (BASELINE config 4's batch shape:
31 250 reads x 2 kb).  No oracle run here (the host aligner would take minutes): parity at this shape is
covered by the size-independent properties in tests/ and by tools/fuzz_parity.py sahlin."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import api, synth  # noqa: E402
from tests.helpers import oracle_sorted_batch  # noqa: E402

n, g, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "sahlin"
rs = synth.generate(n, g, L, 10, 21, seed=3)
t = time.time()
B, view = oracle_sorted_batch(rs)
seqs = [rs.read(int(i))[0] for i in view["orig"]]
off = np.zeros(len(seqs) + 1, np.int64)
off[1:] = np.cumsum([len(x) for x in seqs])
v = dict(view)
v.update(raw_seq=b"".join(seqs), raw_off=off)
print(f"prep {time.time() - t:.1f}s, minimizers {len(view['min_val'])}", flush=True)
ctx = api.Context(0)
p = api.default_params(11, 15, mode)
for r in range(3):
    t = time.time()
    cls, strand, st = ctx.cluster_batch(p, v)
    tm = ctx.timings()
    print(f"hip {1e3 * (time.time() - t):.1f} ms; clusters {st['n_clusters']}, aligned reads {st['n_aln_invoked']}, pairs {st['n_aln_pairs']}, "
          f"rounds {st['aln_rounds']}, resolve iters {st['resolve_iters']}; fwd {tm['ms_align_fwd']:.1f} ms trace {tm['ms_align_trace']:.1f} ms "
          f"cells {tm['n_align_cells']:.3e}", flush=True)
