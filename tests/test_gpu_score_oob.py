"""The scoring kernel's two builds side by side.

k_score_part<PT, true> has no per-posting window test: the counters of its T visible targets sit at the very end of the
workgroup's LDS allocation, so the counter address of every posting the test would reject lies beyond the allocation, and
the variant is right iff gfx950 drops an LDS atomic there.  ioc_ctx_create PROBES that on the device (k_lds_oob_probe: the
kernel's own static LDS layout, two dynamic sizes, every word of [end, end + 256 KB) and of the 1 MB far base hit once by
workgroups that share CUs).  Since round 4 the masked variant k_score_part<PT, false> (defined behaviour) is the DEFAULT — the
other one bought 1.6 % — and IOC_SCORE_OOB=1 asks for the variant without the test, which is then used only if the probe passes.
Here: the probe's verdict is visible in ioc_timings, and the two variants write IDENTICAL candidate lists
(target, strand, Size — the histograms of GetMinimizerHits / ConsolidateMinimizerHits, src/minimizer.cpp:44-76,
src/cluster.cpp:609-615) for every query of config 1 and of short_dup, which also equal the oracle-checked full tables."""
import numpy as np
import pytest

from isonclust2_amd import api, synth
from tests.helpers import oracle_sorted_batch

pytestmark = pytest.mark.gpu


def _scored(view, oob, monkeypatch):
    if oob is None:
        monkeypatch.delenv("IOC_SCORE_OOB", raising=False)
    else:
        monkeypatch.setenv("IOC_SCORE_OOB", oob)
    ctx = api.Context(0)
    p = api.default_params(11, 15, "fast")
    cls, strand, st = ctx.cluster_batch(p, view)
    tm = ctx.timings()
    n = len(view["hpc_len"])
    lists = [ctx.scored_candidates(q, 2 * n + 2) for q in range(n)]
    ctx.close()
    return cls, strand, tm, lists


@pytest.mark.parametrize("cfg,seed", [("config1", 1), ("short_dup", 1)])
def test_both_variants_write_the_same_histograms(cfg, seed, monkeypatch):
    rs = synth.generate_config(cfg, seed=seed)
    _, view = oracle_sorted_batch(rs)
    c0, s0, tm0, l0 = _scored(view, "0", monkeypatch)
    c1, s1, tm1, l1 = _scored(view, "1", monkeypatch)
    assert tm0["score_oob"] == 0 and tm0["score_oob_probe"] == -1          # not asked for: no probe, the masked variant
    assert tm1["score_oob_probe"] >= 0                                        # asked for: the probe ran and decided
    assert tm1["score_oob"] == (1 if tm1["score_oob_probe"] == 0 else 0)
    assert np.array_equal(c0, c1) and np.array_equal(s0, s1)
    total = 0
    for q, ((k0, z0), (k1, z1)) in enumerate(zip(l0, l1)):
        assert np.array_equal(k0, k1) and np.array_equal(z0, z1), q
        total += len(k0)
    assert total > rs.n            # the lists are not empty


def test_default_is_the_masked_variant_and_the_probe_guards_the_other(monkeypatch):
    rs = synth.generate_config("tiny", seed=7)
    _, view = oracle_sorted_batch(rs)
    _, _, tm, _ = _scored(view, None, monkeypatch)
    assert tm["score_oob"] == 0 and tm["score_oob_probe"] == -1         # default: defined behaviour, nothing probed
    _, _, tm, _ = _scored(view, "1", monkeypatch)
    assert tm["score_oob_probe"] >= 0                                 # it ran
    assert tm["score_oob"] == (1 if tm["score_oob_probe"] == 0 else 0)  # and its verdict decides
    if tm["score_oob_probe"] != 0:
        import warnings
        warnings.warn(f"LDS out-of-bounds probe FAILED on this device ({tm['score_oob_probe']}): the masked scoring variant is in use")
