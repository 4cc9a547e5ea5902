"""BASELINE.json configs[4] ("config 5" of SURVEY.md §8(d)) at batch scale: the workload shared by the golden generator
(tools/gen_golden_config5.py, oracle side, build container) and the GPU test (tests/test_gpu_config5.py, product side).

  reads      NB = 4 chunks of PER = 31 250 reads x 2 kb from ONE transcriptome of 1500 transcripts (the chunks of
             tools/cli_config5.py: seeds 1000 + c, transcript seed 11), sorted GLOBALLY by quality score as `isONclust2
             sort` does (src/main.cpp:122) and cut into NB batches of PER consecutive reads (`-B 62500 -M 31250`)
  mode       sahlin, k = 11, w = 15
  consensus  on: ConsMaxSize 150 (`-c 150`), ConsMinSize 20, ConsPeriod 400 (a cluster of this data holds ~ 21 reads:
             the default ConsMinSize of 50 would never fire inside a leaf)
  graphs     tests/helpers.py::ToyGraphs on both sides (spoa is absent from the reference tree: parity of the graphs is
             unpinned; what is pinned is everything around them — which joins take a consensus, the weighted error rates,
             the re-minimized representatives, UpdateMinDB, ConsPurge, and the merges' use of the right graphs' sizes)
  tree       leaves b0..b3, then (b0 + b1), (b2 + b3), then ((b0 + b1) + (b2 + b3)): src/cluster.cpp:67-322 with two batches

Digests: FNV-1a over (cluster, strand) of reads 0..NB*PER-1 in generator order (-1 / 0 = not in the batch), sha256 over
the graph-operation log, sha256 over the MinDB (keys, offsets, postings)."""
import hashlib

import numpy as np

from isonclust2_amd import synth

NB, PER, G, LEN, TR_SEED = 4, 31250, 1500, 2000, 11
K, W, MODE = 11, 15, "sahlin"
CONS_MIN, CONS_MAX, CONS_PERIOD = 20, 150, 400
TREE = [(0, 1), (2, 3), (0, 2)]          # (left slot, right slot): the result replaces the left slot


def concat_readsets(parts):
    offs = [np.zeros(1, np.int64)]
    base = 0
    for p in parts:
        offs.append(p.offs[1:] + base)
        base += int(p.offs[-1])
    return synth.ReadSet(seq=np.concatenate([p.seq for p in parts]), qual=np.concatenate([p.qual for p in parts]),
                         offs=np.concatenate(offs), transcript=np.concatenate([p.transcript for p in parts]),
                         strand=np.concatenate([p.strand for p in parts]), tag=f"config5[{len(parts)} chunks]")


def reads(nb=NB, per=PER):
    return concat_readsets([synth.generate(per, G, LEN, 10, 21, seed=1000 + c, tr_seed=TR_SEED) for c in range(nb)])


def log_sha(log):
    return hashlib.sha256(repr(log).encode()).hexdigest()[:32]


def mindb_sha(keys, offs, post):
    h = hashlib.sha256()
    for a, t in ((keys, np.uint32), (offs, np.int64), (post, np.uint32)):
        h.update(np.ascontiguousarray(a, t).tobytes())
    return h.hexdigest()[:32]


# ---- the same branch with REAL partial-order graphs -------------------------------------------------------
# One leaf pair and their merge at a scale the oracle's scalar POA finishes in minutes in the build container: REAL_NB batches of
# REAL_PER reads x 2 kb over REAL_G transcripts (50 reads per cluster: every join between ConsMinSize 20 and ConsMaxSize 150 takes
# a consensus), sahlin, the graphs = the oracle's POA (oracle/poa_oracle.cpp) on the golden's side, the product's engine
# (ioc_poa.hip) on the test's.  Beyond the toy-graph record's digests: the consensus strings IN EVENT ORDER (cluster id + sha256
# of the string, one entry per consensus event), and at the end every cluster's graph (letters, topological ranks, weighted
# edges) and consensus.  spoa itself stays unpinned: this is parity with the oracle's restatement of it.
REAL_NB, REAL_PER, REAL_G = 2, 3000, 60


def real_reads(nb=REAL_NB, per=REAL_PER, g=REAL_G):
    return concat_readsets([synth.generate(per, g, LEN, 10, 21, seed=2000 + c, tr_seed=TR_SEED + 1) for c in range(nb)])


def events_sha(events):
    """events: [(cluster id, consensus bytes)] in event order."""
    h = hashlib.sha256()
    for c, s in events:
        h.update(f"{int(c)}:".encode() + hashlib.sha256(s).digest())
    return h.hexdigest()[:32]


def graphs_sha(store, n_clusters, side=0):
    """Every cluster's graph and consensus at the end: store.graph(c) -> (bases, rank, edge_from, edge_to, edge_w), store.consensus(c)."""
    hg, hc = hashlib.sha256(), hashlib.sha256()
    for c in range(n_clusters):
        bases, rank, ef, et, ew = store.graph(c, side)[:5]
        hg.update(bytes(bases) + np.asarray(rank, np.int32).tobytes())
        e = sorted(zip(np.asarray(ef).tolist(), np.asarray(et).tolist(), np.asarray(ew).tolist()))
        hg.update(np.asarray(e, np.int64).tobytes())
        hc.update(hashlib.sha256(store.consensus(c, side)).digest())
    return hg.hexdigest()[:32], hc.hexdigest()[:32]
