#!/usr/bin/env python3
"""Developer run of consensus mode at scale: a synthetic batch through ioc_cluster_consensus with the product's POA
engine behind the graph operations (CLI defaults ConsMinSize 20, ConsMaxSize 100, ConsPeriod 400)."""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from isonclust2_amd import _lib, api, pipeline, synth  # noqa: E402
from tests.test_gpu_poa import Poa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="600,30,4000", help="n_reads,n_transcripts,length")
ap.add_argument("--mode", default="fast")
ap.add_argument("--cons", default="20,100,400", help="ConsMinSize,ConsMaxSize,ConsPeriod")
a = ap.parse_args()
n, g, ln = (int(x) for x in a.shape.split(","))
cmin, cmax, per = (int(x) for x in a.cons.split(","))
rs = synth.generate(n, g, ln, 10, 21, seed=1)
ctx = api.Context(0)
sb, _ = pipeline.sort_stage(ctx, rs, 11, 15)     # the product's own GPU sort stage
v = sb.view
poa = Poa(ctx)
cargs = _lib.ConsensusArgs(cons_min_size=cmin, cons_max_size=cmax, cons_period=per, left_depth=-1, left_sizes=None)
t = time.time()
cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, a.mode), None, v, cargs, poa.ops)
dt = time.time() - t
t = time.time()
poa.graph(0)      # (a .cer writer would save every graph now: works off whatever additions are still queued)
dt_flush = time.time() - t
print(f"{rs.tag} {a.mode}: {dt:.2f} s + {dt_flush:.2f} s for the queued graph additions ({rs.n / (dt + dt_flush):.0f} reads/s); {st}")
poa.close()
