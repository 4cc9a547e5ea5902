"""GPU parity of sahlin / furious mode (minimizer mapping + alignment fallback, src/cluster.cpp:545-566)
against the oracle.  parasail is absent from the reference tree: the oracle aligns with its OWN scalar aligner
(oracle.cpp sg_trace, pinned by the reference's AlnRatioTest vector), the product aligns on the GPU
(ioc_align_pairs) — two independent implementations.  Checked: which reads reach the fallback, which candidates are
aligned in which order, every verdict, and how the batched, speculative verdicts feed back into the greedy loop."""
import ctypes as C

import numpy as np
import pytest

from isonclust2_amd import _lib, api, synth
from oracle import pyoracle as po
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def hooked_oracle():
    """Since round 2 the oracle aligns with its OWN scalar aligner (oracle.cpp sg_trace): nothing of the product sits
    behind the oracle in these tests (the name of the fixture is historical)."""
    po.lib().orc_set_aligner(None)
    po.lib().orc_use_builtin_aligner(1)
    yield None


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _with_sequences(rs, view):
    seqs = [rs.read(int(i))[0] for i in view["orig"]]
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    v = dict(view)
    v.update(raw_seq=b"".join(seqs), raw_off=off)
    return v


@pytest.mark.parametrize("mode", ["sahlin", "furious"])
@pytest.mark.parametrize("seed", [1, 2])
def test_alignment_fallback_modes(ctx, hooked_oracle, mode, seed):
    rs = synth.generate(220, 25, 500, 9, 20, seed=seed)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, ostat = oracle_entry_assignments(B, view, mode=mode)
    cls, strand, st = ctx.cluster_batch(api.default_params(11, 15, mode), _with_sequences(rs, view))
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, (len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
    assert st["n_aln_invoked"] == ostat["aln_invoked"] > 0
    assert st["n_clusters"] == B.n_clusters()


def test_host_aligner_route(ctx, hooked_oracle, monkeypatch):
    """IOC_ALIGN_HOST=1: same driver, pairs aligned by the host aligner one verdict at a time."""
    monkeypatch.setenv("IOC_ALIGN_HOST", "1")
    rs = synth.generate(120, 15, 400, 9, 20, seed=3)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, ostat = oracle_entry_assignments(B, view, mode="sahlin")
    cls, strand, st = ctx.cluster_batch(api.default_params(11, 15, "sahlin"), _with_sequences(rs, view))
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)
    assert st["n_aln_invoked"] == ostat["aln_invoked"] > 0


@pytest.mark.parametrize("mode", ["sahlin", "furious"])
def test_duplicated_transcripts_order_dependent_verdicts(ctx, hooked_oracle, mode):
    """Paralog-like duplicates: several candidates tie at the top Size and more than one aligns, so the
    verdict depends on the reference's candidate order (cluster.cpp:481-511) and on the cluster numbering."""
    rs = synth.generate(400, 40, 300, 9, 20, seed=7, dup_every=2)
    B, view = oracle_sorted_batch(rs)
    ocl, ost, ostat = oracle_entry_assignments(B, view, mode=mode)
    cls, strand, st = ctx.cluster_batch(api.default_params(11, 15, mode), _with_sequences(rs, view))
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, (len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
    assert st["n_aln_invoked"] == ostat["aln_invoked"] > 0


def test_sahlin_needs_sequences(ctx):
    rs = synth.generate_config("tiny")
    B, view = oracle_sorted_batch(rs)
    with pytest.raises(api.IocError):
        ctx.cluster_batch(api.default_params(11, 15, "sahlin"), view)
