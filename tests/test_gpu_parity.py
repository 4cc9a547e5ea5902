"""GPU parity: the HIP path, driven through the C ABI, must give cluster assignments bit-identical
to the CPU oracle on the same sorted batch (BASELINE.json north_star)."""
import numpy as np
import pytest

from isonclust2_amd import api, synth
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _check(ctx, rs, k=11, w=15):
    B, view = oracle_sorted_batch(rs, k, w)
    ocl, ost, ostat = oracle_entry_assignments(B, view)
    cls, strand, st = ctx.cluster_batch(api.default_params(k, w, "fast"), view)
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, f"{len(bad)} differing assignments, first at entry {bad[:5]}: hip={cls[bad[:5]]},{strand[bad[:5]]} oracle={ocl[bad[:5]]},{ost[bad[:5]]}"
    assert st["n_clusters"] == B.n_clusters()
    return st, ostat


def test_tiny(ctx):
    _check(ctx, synth.generate_config("tiny"))


def test_config1_500_reads(ctx):
    st, ostat = _check(ctx, synth.generate_config("config1"))
    assert st["n_joined"] == ostat["joins"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_short_reads_with_duplicated_transcripts_ties(ctx, seed):
    """Short reads over duplicated transcripts: >= 2 passing candidates tie at the winning Size, so the
    winner is defined by libstdc++'s unordered_map/std::sort order (SURVEY §7 hard part 2)."""
    st, ostat = _check(ctx, synth.generate_config("short_dup", seed=seed))
    assert st["n_tie_replays"] >= ostat["tie_reads"]


def test_k13_w20(ctx):
    _check(ctx, synth.generate(300, 30, 1200, 10, 21, seed=5), k=13, w=20)


def test_mixed_lengths(ctx):
    _check(ctx, synth.generate(400, 40, 900, 9, 22, seed=9, len_jitter=0.6))


def test_long_reads_multi_pass_evaluation(ctx):
    """40 kb reads: > 4096 minimizers per strand and > 4096 distinct values per representative, i.e. the
    multi-pass evaluator and the large-set (global binary search) membership path."""
    _check(ctx, synth.generate(60, 6, 40000, 11, 20, seed=13))


def test_range_passes_and_unpartitioned_kernel(ctx, monkeypatch):
    """More visible targets than the LDS histogram window: k_score_t with range passes (forced small)."""
    monkeypatch.setenv("IOC_SCORE_RANGE", "96")
    _check(ctx, synth.generate_config("config1", seed=5))
    monkeypatch.setenv("IOC_SCORE_RANGE", "8192")
    monkeypatch.setenv("IOC_SCORE_PARTS", "0")
    _check(ctx, synth.generate_config("short_dup", seed=4))


def test_u32_partial_histograms(ctx, monkeypatch):
    """Strands of >= 65536 minimizers switch the partial histograms from packed u16 to u32 (forced here)."""
    monkeypatch.setenv("IOC_PART32", "1")
    _check(ctx, synth.generate_config("config1", seed=8))


def test_tiny_work_queue_forces_repeated_sweeps(ctx, monkeypatch):
    """A work queue smaller than the number of pending evaluations: sweeps repeat until complete."""
    monkeypatch.setenv("IOC_QUEUE_CAP", "64")
    _check(ctx, synth.generate_config("config1", seed=6))


def test_empty_and_degenerate_batches(ctx):
    p = api.default_params(11, 15, "fast")
    empty = dict(off_fwd=np.zeros(1, np.int64), off_rev=np.zeros(1, np.int64), min_val=np.zeros(0, np.uint32),
                 min_pos=np.zeros(0, np.uint32), raw_len=np.zeros(0, np.uint32), hpc_len=np.zeros(0, np.uint32),
                 score=np.zeros(0), raw_err=np.zeros(0), hpc_err=np.zeros(0), state=np.zeros(0, np.uint8))
    cls, strand, st = ctx.cluster_batch(p, empty)
    assert len(cls) == 0 and st["n_clusters"] == 0
    # one read; and a batch in which every read is gated (quality below MinQual)
    rs = synth.generate(1, 1, 800, 15, 15, seed=1)
    _check(ctx, rs)
    rs = synth.generate(20, 2, 600, 3, 5, seed=2)
    B, view = oracle_sorted_batch(rs)
    cls, strand, st = ctx.cluster_batch(p, view)
    ocl, ost, _ = oracle_entry_assignments(B, view)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
    assert st["n_gated"] == int((ocl < 0).sum())


def test_mode_none_opens_one_cluster_per_read(ctx):
    """`cluster` without -x: Mode = None, no branch of getBestCluster fires (cluster.cpp:545-567)."""
    rs = synth.generate_config("tiny")
    B, view = oracle_sorted_batch(rs)
    ocl, ost, _ = oracle_entry_assignments(B, view, mode="none")
    cls, strand, st = ctx.cluster_batch(api.default_params(11, 15, "none"), view)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)


def test_every_prefix_of_a_batch(ctx):
    """The greedy loop on the first m reads for every m: batches that END in a join, in a new cluster, in a
    read whose best candidates fail.  (The resolve's lazy stage used to end without its exact sweeps when the
    LAST query was the first one whose cluster flag changed; the consensus driver's short re-runs found it and
    tests/test_consensus.py::test_device_consensus_equals_oracle[fast-shape1-...] is its regression test.)"""
    rs = synth.generate(140, 7, 600, 11, 21, seed=9)
    B, view = oracle_sorted_batch(rs)
    order = view["orig"]
    p = api.default_params(11, 15, "fast")
    joins_at_end = 0
    for m in range(1, rs.n + 1):
        sub = rs.subset(order[:m])                     # the m best reads, already in sorted order
        Bm, vm = oracle_sorted_batch(sub)
        assert np.array_equal(vm["orig"], np.arange(m))
        ocl, ost, _ = oracle_entry_assignments(Bm, vm)
        cls, strand, st = ctx.cluster_batch(p, vm)
        assert np.array_equal(cls, ocl) and np.array_equal(strand, ost), m
        joins_at_end += int(m > 1 and cls[m - 1] < cls[:m].max())
    assert joins_at_end > 20


def _concat(a, b):
    return synth.ReadSet(seq=np.concatenate([a.seq, b.seq]), qual=np.concatenate([a.qual, b.qual]),
                         offs=np.concatenate([a.offs, a.offs[-1] + b.offs[1:]]), transcript=np.concatenate([a.transcript, a.transcript.max() + 1 + b.transcript]),
                         strand=np.concatenate([a.strand, b.strand]), tag=a.tag + "+" + b.tag)


@pytest.mark.parametrize("mode", ["fast", "sahlin"])
def test_ultra_long_read(ctx, mode):
    """A read of 200 kb (50 000 forward minimizers: beyond every in-LDS structure of the index build) among 200 ordinary ones must
    not stop the batch — the reference has no such limit (plain vectors, minimizer.cpp:78-123; round 4 returned IOC_ERR_CAPACITY
    above 32 768 forward minimizers).  Fast mode: two reads of one 200 kb transcript (the second is evaluated against the first
    one's 45 000-value set); sahlin: one (the oracle's scalar aligner would need minutes for a 200 kb x 200 kb pair).  The GPU
    sort stage must extract the long read's minimizers like the oracle's, too."""
    from isonclust2_amd import pipeline
    rs = _concat(synth.generate(200, 20, 1500, 10, 21, seed=71), synth.generate(2 if mode == "fast" else 1, 1, 200000, 12, 18, seed=72))
    B, view = oracle_sorted_batch(rs)
    assert int(np.diff(view["off_fwd"]).max()) > 40000
    ocl, ost, _ = oracle_entry_assignments(B, view, mode=mode)
    v = dict(view)
    if mode != "fast":
        seqs = [rs.read(int(i))[0] for i in view["orig"]]
        off = np.zeros(len(seqs) + 1, np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        v.update(raw_seq=b"".join(seqs), raw_off=off)
    cls, strand, st = ctx.cluster_batch(api.default_params(11, 15, mode), v)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
    if mode == "fast":
        long_entries = np.nonzero(np.diff(view["off_fwd"]) > 40000)[0]
        assert len(long_entries) == 2 and cls[long_entries[0]] == cls[long_entries[1]]   # (the second long read joined the first)
    sb, order = pipeline.sort_stage(ctx, rs, 11, 15)
    assert np.array_equal(order, view["orig"])
    for key in ("off_fwd", "off_rev", "min_val", "min_pos", "hpc_len"):
        assert np.array_equal(np.asarray(sb.view[key]), np.asarray(view[key])), key
