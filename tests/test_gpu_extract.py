"""GPU parity of the sort-stage kernels (K1: HPC + k-mer encode + minimizers on both strands,
K2: quality score / error rate) against the CPU oracle.  Integer outputs and fp64 outputs must both
be bit-exact (the fp64 recurrences keep the reference's operation order, no FMA)."""
import numpy as np
import pytest

from isonclust2_amd import api, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _reads(rs):
    return [rs.read(i) for i in range(rs.n)]


@pytest.mark.parametrize("k", [3, 11, 13])
def test_qual_scores_bitwise(ctx, k):
    rs = synth.generate(200, 20, 700, 5, 30, seed=4, len_jitter=0.9)
    score, err = ctx.qual_scores(rs.offs, rs.qual, k)
    for i, (s, q) in enumerate(_reads(rs)):
        if len(s) > 2 * k:
            qs = po.qual_score(q, k)
            exp = qs if qs > 0 else -1.0
            assert score[i] == exp, (i, score[i], exp)
            assert err[i] == po.error_rate(q, nomin=True)
        else:
            assert score[i] == -1.0 and err[i] == 1.0


def test_qual_scores_reference_kat(ctx, kat):
    g = kat["sorting"]
    seqs = [r["seq"].encode() for r in g["reads"]]
    quals = [r["qual"].encode() for r in g["reads"]]
    offs = np.zeros(len(seqs) + 1, np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    score, _ = ctx.qual_scores(offs, np.frombuffer(b"".join(quals), np.uint8), g["k"])
    order = np.argsort(-score, kind="stable")
    assert [g["reads"][i]["name"] for i in order] == g["expected_order"]


@pytest.mark.parametrize("k,w", [(11, 15), (13, 20), (10, 10), (17, 24), (20, 21)])
def test_extract_minimizers_bitwise(ctx, k, w):
    rs = synth.generate(150, 15, 900, 8, 25, seed=k + w, len_jitter=0.8)
    ex = ctx.extract_minimizers(rs.offs, rs.seq, rs.qual, k, w)
    total = int(ex["off_rev"][-1])
    mn, ps = ctx.extracted_download(total)
    for i, (s, q) in enumerate(_reads(rs)):
        hs, hq = po.hpc(s, q)
        assert ex["hpc_len"][i] == len(hs)
        if len(hs) < 2 * k or len(hs) < w:
            assert ex["status"][i] == 1
            assert ex["off_fwd"][i + 1] == ex["off_fwd"][i] and ex["off_rev"][i + 1] == ex["off_rev"][i]
            continue
        assert ex["status"][i] == 0
        assert ex["hpc_err"][i] == po.error_rate(hq, nomin=True)
        for strand, off in ((0, ex["off_fwd"]), (1, ex["off_rev"])):
            seq = hs if strand == 0 else po.revcomp(hs)
            emn, eps, eix = po.minimizers(po.kmer_encode(seq, k), k, w)
            a, b = int(off[i]), int(off[i + 1])
            assert b - a == len(emn), (i, strand, b - a, len(emn))
            assert np.array_equal(mn[a:b], emn)
            assert np.array_equal(ps[a:b], eps)


def test_extract_reference_kats(ctx, kat):
    g = kat["hpc"]
    seq, qual = g["seq"].encode(), g["qual"].encode()
    offs = np.array([0, len(seq)], np.int64)
    ex = ctx.extract_minimizers(offs, np.frombuffer(seq, np.uint8), np.frombuffer(qual, np.uint8), 2, 4)
    assert ex["hpc_len"][0] == len(g["expected_seq"])
    m = kat["minimizer"]  # "ACGCCGATC" is already homopolymer-free except CC -> use its HPC form via oracle
    hs, _ = po.hpc(m["seq"].encode(), b"I" * len(m["seq"]))
    emn, eps, _ = po.minimizers(po.kmer_encode(hs, m["k"]), m["k"], m["w"])
    offs = np.array([0, len(m["seq"])], np.int64)
    ex = ctx.extract_minimizers(offs, np.frombuffer(m["seq"].encode(), np.uint8),
                                np.frombuffer(b"I" * len(m["seq"]), np.uint8), m["k"], m["w"])
    mn, ps = ctx.extracted_download(int(ex["off_rev"][-1]))
    a, b = int(ex["off_fwd"][0]), int(ex["off_fwd"][1])
    assert np.array_equal(mn[a:b], emn) and np.array_equal(ps[a:b], eps)


def test_extract_edge_cases(ctx):
    k, w = 11, 15
    reads = [b"A" * 500,                       # one homopolymer run -> HPC length 1
             b"ACGT" * 6,                       # HPC length 24 >= 2k, >= w
             b"ACGTACGTACGTACGTACGTAC",         # HPC length 22 == 2k
             b"ACGTACGTACGTACGTACGTA",          # 21 < 2k
             b"ACGTNACGT" * 10,                 # non-ACGT
             b"AC" * 40]
    quals = [bytes([40 + (i * 7 + j) % 50 for j in range(len(r))]) for i, r in enumerate(reads)]
    offs = np.zeros(len(reads) + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    ex = ctx.extract_minimizers(offs, np.frombuffer(b"".join(reads), np.uint8),
                                np.frombuffer(b"".join(quals), np.uint8), k, w)
    assert list(ex["status"]) == [1, 0, 0, 1, 2, 0]
    mn, ps = ctx.extracted_download(int(ex["off_rev"][-1]))
    for i in (1, 2, 5):
        hs, hq = po.hpc(reads[i], quals[i])
        for strand, off in ((0, ex["off_fwd"]), (1, ex["off_rev"])):
            seq = hs if strand == 0 else po.revcomp(hs)
            emn, eps, _ = po.minimizers(po.kmer_encode(seq, k), k, w)
            a, b = int(off[i]), int(off[i + 1])
            assert np.array_equal(mn[a:b], emn) and np.array_equal(ps[a:b], eps)


def test_extract_then_cluster_matches_oracle(ctx):
    """End-to-end without the oracle on the product side: raw reads -> GPU sort stage -> GPU clustering,
    compared with the oracle's sort + cluster on the same reads."""
    from tests.helpers import oracle_entry_assignments, oracle_sorted_batch
    rs = synth.generate_config("config1", seed=3)
    k, w = 11, 15
    B, view = oracle_sorted_batch(rs, k, w)
    ocl, ost, _ = oracle_entry_assignments(B, view)
    # product side: scores -> stable sort -> extraction in sorted order -> cluster
    score, err = ctx.qual_scores(rs.offs, rs.qual, k)
    order = np.argsort(-score, kind="stable")
    assert np.array_equal(order, view["orig"])
    lens = np.diff(rs.offs)[order]
    so = np.zeros(rs.n + 1, np.int64)
    so[1:] = np.cumsum(lens)
    idx = np.concatenate([np.arange(rs.offs[i], rs.offs[i + 1]) for i in order])
    ex = ctx.extract_minimizers(so, rs.seq[idx], rs.qual[idx], k, w)
    p = api.default_params(k, w, "fast")
    ctx.set_params(p)
    keep = (ex["status"] == 0) & (score[order] >= 0) & (-10 * np.log10(err[order]) > 7.0)
    cell = np.array([api.host_err_cell(e) if kp else 1 for e, kp in zip(ex["hpc_err"], keep)], np.uint8)
    need = np.array([api.host_min_total(h, p.mapped_threshold) if kp else 0xFFFFFFFE
                     for h, kp in zip(ex["hpc_len"], keep)], np.uint32)
    ctx.queries_from_extracted(keep, cell, need)
    ctx.left_load(0, None, None, None, None)
    cls, strand, st = ctx.cluster_resident()
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
