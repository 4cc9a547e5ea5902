"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/isonclust2_hip.h
declares, refuses to run without a GPU (no CPU fallback), and its pure-host helpers agree with the
oracle's restatement of the reference (p_emp_prob.cpp / util.cpp / cluster.cpp:390-400)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from isonclust2_amd import _lib, api
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "isonclust2_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ioc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(api.IocError) as e:
        api.Context(0)
    assert e.value.code == -6  # IOC_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "isonclust2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "liboracle" not in src and "oracle/" not in src.replace("oracle/_ref", ""), f


@pytest.mark.parametrize("k,w", [(11, 15), (13, 20), (10, 10), (30, 100), (20, 27)])
def test_gap_limits_equal_oracle(k, w):
    g, p = api.host_gap_limits(k, w, 0.1)
    tab, filled = po.pmin_table(k, w)
    assert filled == 225
    assert np.array_equal(p, tab)
    for a in range(15):
        for b in range(15):
            assert g[a, b] == po.lib().orc_gap_limit(tab[a, b], 0.1)


def test_gap_limits_outside_table_fail_like_the_reference():
    with pytest.raises(api.IocError):
        api.host_gap_limits(31, 40)   # k = 31 passes arg checks but has no rows (p_emp_prob.cpp:87-89)
    with pytest.raises(api.IocError):
        api.host_gap_limits(11, 200)


def test_err_cell_equals_reference_rounding():
    tab, _ = po.pmin_table(11, 15)
    rng = np.random.default_rng(5)
    xs = list(rng.random(3000) * 0.25) + [0.0, 0.004999, 0.005, 0.0149999, 0.015, 0.145, 0.1549, 0.155, 0.2, 1.0]
    for e in xs:
        c = api.host_err_cell(e)
        assert 1 <= c <= 15
        # the oracle's GetPMinShared picks the same cell
        assert po.pmin_lookup(tab, e, 0.05) == tab[c - 1, 4]
    assert api.host_err_cell(float("nan")) == 0


def test_min_total_is_the_float_threshold():
    rng = np.random.default_rng(6)
    for L in list(rng.integers(20, 200000, 400)) + [22, 12500, 100, 101]:
        t = api.host_min_total(int(L), 0.65)
        f = lambda T: float(np.float32(np.float64(T) / np.float64(L))) >= 0.65  # float widened to double, as in C++
        assert f(t) and (t == 0 or not f(t - 1)), (L, t)
    assert api.host_min_total(0, 0.65) == 0xFFFFFFFE
    assert api.host_min_total(100, 1.5) == 0xFFFFFFFE
