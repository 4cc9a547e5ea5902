"""The oracle's scalar POA (oracle/poa_oracle.cpp) checked on the CPU against what can be stated without it: the plain-Python
recurrence of the local sequence-to-graph DP with the convex gap of src/main.cpp:285-290, the validity and score of the
alignment path it returns, the topological order, edge weights as sums of per-base weights, the purge of src/consensus.cpp:128-137,
and that the heaviest-bundle consensus of noisy copies is close to their source.  (spoa is absent from the reference tree: this
pins the oracle's internal consistency, not spoa.)"""
import random

from oracle import pyoracle as po
from tests.poa_common import _path_score, _ref_score, mutate, random_addition


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def test_alignment_score_equals_the_plain_recurrence_and_the_path_scores_the_same():
    rng = random.Random(5)
    adds = 0
    for g in range(12):
        P = po.OraclePoa()
        truth = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([40, 90, 200])))
        P.create(0, mutate(rng, truth, 0.05))
        for t in range(rng.randint(4, 8)):
            r = random_addition(rng, truth, t)
            if not r:
                continue
            bases, rank, ef, et, ew = P.graph(0)
            want = _ref_score(bases, rank, ef, et, r)
            P.add(0, r, w=1 + t % 3)
            nodes, pos, score = P.last_alignment()
            adds += 1
            assert score == want, (g, t)
            assert _path_score(bases, ef, et, r, nodes, pos) == score, (g, t)
            bases, rank, ef, et, ew = P.graph(0)
            order = {int(v): i for i, v in enumerate(rank)}
            assert sorted(rank.tolist()) == list(range(len(bases)))
            assert all(order[int(a)] < order[int(b)] for a, b in zip(ef, et))
        P.close()
    assert adds > 50


def test_weights_aligned_columns_and_purge():
    P = po.OraclePoa()
    P.create(0, b"ACGTACGTAC")
    bases, rank, ef, et, ew = P.graph(0)
    assert bases == b"ACGTACGTAC" and ew.tolist() == [2] * 9 and P.size(0) == 1       # weight[i - 1] + weight[i] per edge
    P.add(0, b"ACGTACGTAC", w=3)                                                          # same path: weights grow, no new node
    bases, rank, ef, et, ew = P.graph(0)
    assert len(bases) == 10 and ew.tolist() == [8] * 9 and P.size(0) == 2
    P.add(0, b"ACGTTCGTAC", w=1)                                                          # one mismatch: an aligned node
    bases, rank, ef, et, ew, al = P.graph(0, aligned=True)
    assert len(bases) == 11 and bases[10:11] == b"T" and al[4] == [10] and al[10] == [4]
    order = rank.tolist()
    assert abs(order.index(4) - order.index(10)) == 1                                     # a column's nodes are neighbours in the order
    assert P.consensus(0) == b"ACGTACGTAC"
    P.purge(0, b"ACGTACGTAC", w=3)                                                        # ConsPurge: one sequence, the old count as weight
    bases, rank, ef, et, ew = P.graph(0)
    assert P.size(0) == 1 and ew.tolist() == [6] * 9
    P.add(0, b"GGGGGGGG")                                                                 # unrelated: best local alignment is one base
    assert P.size(0) == 2
    P.close()


def test_consensus_of_noisy_copies_is_close_to_the_source():
    rng = random.Random(3)
    P = po.OraclePoa()
    truth = bytes(rng.choice(b"ACGT") for _ in range(500))
    reads = [mutate(rng, truth, 0.12) for _ in range(14)]
    P.create(5, reads[0])
    for r in reads[1:]:
        P.add(5, r)
    d0, d = _edit_distance(reads[0], truth), _edit_distance(P.consensus(5), truth)
    assert d <= 0.03 * len(truth) and d < d0 / 3, (d, d0)
    P.close()
