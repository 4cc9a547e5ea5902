"""The product's POA engine (isonclust2_amd/csrc/ioc_poa.hip: sequence-to-graph DP and traceback on the GPU, graphs and
heaviest-bundle consensus on the host) against the ORACLE's scalar POA (oracle/poa_oracle.cpp: an independent restatement
of the published algorithm in the shape of spoa 4.0's scalar engine and graph — src/consensus.cpp:15-32, 91, 128-137,
src/main.cpp:285-324).  Both stores are fed the same operations; compared after EVERY addition: the alignment (score and
every (node, position) pair) and, per graph, the nodes' letters, the edges with their weights, the topological order and the
consensus string.  "Parity with the oracle's POA; spoa unpinned" (its source is absent from the reference tree)."""
import ctypes as C
import random

import numpy as np
import pytest

from isonclust2_amd import api
from oracle import pyoracle as po
from tests.poa_common import mutate, random_addition
from tests.test_gpu_poa import Poa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _edges(ef, et, ew):
    return sorted(zip(ef.tolist(), et.tolist(), ew.tolist()))


def _same_graph(dev, orc, idx, tag):
    db, dr, def_, det, dew = dev.graph(idx)
    ob, orr, oef, oet, oew = orc.graph(idx)
    assert db == ob, (tag, "letters")
    assert _edges(def_, det, dew) == _edges(oef, oet, oew), (tag, "edges / weights")
    assert dr.tolist() == orr.tolist(), (tag, "topological order")
    assert dev.size(idx) == orc.size(idx), (tag, "sequences")
    assert dev.consensus(idx) == orc.consensus(idx), (tag, "consensus")


def _same_alignment(dev, orc, tag):
    dn, dp, ds = dev.last_alignment()
    on, op, os_ = orc.last_alignment()
    assert ds == os_, (tag, "score", ds, os_)
    first = next((x for x in range(min(len(dn), len(on))) if dn[x] != on[x] or dp[x] != op[x]), None)
    assert first is None and len(dn) == len(on), (tag, "alignment differs at pair", first, len(dn), len(on),
                                                  None if first is None else (dn[first - 2:first + 3].tolist(), dp[first - 2:first + 3].tolist(),
                                                                              on[first - 2:first + 3].tolist(), op[first - 2:first + 3].tolist()))


def test_random_additions_give_the_oracles_graphs(ctx):
    rng = random.Random(41)
    adds = 0
    for g in range(34):
        dev, orc = Poa(ctx), po.OraclePoa()
        ln = rng.choice([40, 90, 200, 260, 330, 700])
        truth = bytes(rng.choice(b"ACGT") for _ in range(ln))
        first = mutate(rng, truth, rng.choice([0.0, 0.05, 0.15]))
        dev.create(0, first)
        orc.create(0, first)
        _same_graph(dev, orc, 0, (g, "seed"))
        for t in range(rng.randint(4, 9)):
            r = random_addition(rng, truth, t)
            if not r:
                continue
            w = 1 + t % 3
            dev.add(0, r, w=w)
            orc.add(0, r, w=w)
            adds += 1
            _same_alignment(dev, orc, (g, t))
            _same_graph(dev, orc, 0, (g, t))
        dev.close()
        orc.close()
    assert adds >= 200


def test_low_complexity_and_tie_heavy_reads(ctx):
    """Repeats and two-letter sequences: many alignments of equal score, so the tie rules (first maximum, diagonal before the
    gaps, first predecessor, extension before opening) decide the graph."""
    rng = random.Random(43)
    for g in range(12):
        dev, orc = Poa(ctx), po.OraclePoa()
        unit = bytes(rng.choice(b"AC") for _ in range(rng.choice([2, 3, 5])))
        truth = (unit * 80)[: rng.choice([60, 150, 240])]
        dev.create(0, truth)
        orc.create(0, truth)
        for t in range(6):
            r = mutate(rng, truth, 0.08) if t % 2 else (unit * 80)[: rng.randint(20, len(truth))]
            dev.add(0, r, w=1 + t % 2)
            orc.add(0, r, w=1 + t % 2)
            _same_alignment(dev, orc, (g, t))
            _same_graph(dev, orc, 0, (g, t))
        dev.close()
        orc.close()


def test_purge_and_unrelated_reads(ctx):
    """ConsPurge (src/consensus.cpp:128-137) restarts a graph from the representative with the old count as weight; a read that
    shares nothing with the graph has score 0 and is added as a chain of its own."""
    rng = random.Random(47)
    dev, orc = Poa(ctx), po.OraclePoa()
    truth = bytes(rng.choice(b"ACGT") for _ in range(300))
    dev.create(3, truth)
    orc.create(3, truth)
    for t in range(5):
        r = mutate(rng, truth, 0.1)
        dev.add(3, r)
        orc.add(3, r)
    cons = orc.consensus(3)
    assert dev.consensus(3) == cons
    assert dev.ops.purge(dev.ops.user, 0, 3, C.cast(C.c_char_p(cons), C.POINTER(C.c_char)), len(cons), 6) == 0
    orc.purge(3, cons, w=6)
    _same_graph(dev, orc, 3, "purged")
    other = b"T" * 40            # nothing in common with anything at m 4 / n -8? (single matches score 4: local alignment of one base)
    dev.add(3, other)
    orc.add(3, other)
    _same_alignment(dev, orc, "unrelated")
    _same_graph(dev, orc, 3, "unrelated")
    dev.close()
    orc.close()


def test_reads_longer_than_a_tile_column(ctx):
    """2.1 - 2.6 kb reads (row widths of every residue mod 4: the cell arrays' pitch is padded) against graphs 35+ tile rows
    deep: the row carries (prefix maxima, strict-maximum flags, boundary H) cross waves AND column tiles, deletions of 30 - 80
    bases put edges across tile rows, a low-complexity stretch sits on the tile edge at column 1024."""
    rng = random.Random(53)
    for g, ln in enumerate([2101, 2302, 2563, 2048]):
        dev, orc = Poa(ctx), po.OraclePoa()
        truth = bytearray(rng.choice(b"ACGT") for _ in range(ln))
        truth[1000:1050] = (b"AC" * 25)
        truth = bytes(truth)
        first = mutate(rng, truth, 0.05)
        dev.create(0, first)
        orc.create(0, first)
        for t in range(5):
            r = mutate(rng, truth, rng.choice([0.03, 0.1]))
            if t == 2:
                a = rng.randint(900, 1100)
                r = r[:a] + r[a + rng.randint(30, 80):]
            if t == 3:
                r = r[rng.randint(0, 300): len(r) - rng.randint(0, 300)]
            dev.add(0, r, w=1 + t % 2)
            orc.add(0, r, w=1 + t % 2)
            _same_alignment(dev, orc, (g, t))
            _same_graph(dev, orc, 0, (g, t))
        dev.close()
        orc.close()
