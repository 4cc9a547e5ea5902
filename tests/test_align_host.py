"""CPU tests of the host alignment fallback (pure host code of the C-ABI library): the reference's
only alignment known answer (AlnRatioTest, test/isONclust2_test.cpp:137-181) and structural properties
of the semi-global aligner."""
import ctypes as C

import numpy as np

from isonclust2_amd import _lib
from oracle import pyoracle as po


def align(q: bytes, r: bytes, go=3, ge=1, match=2, mismatch=-2):
    L = _lib.load()
    comp = C.create_string_buffer(len(q) + len(r) + 2)
    sc = C.c_int32(0)
    n = L.ioc_host_align(q, len(q), r, len(r), match, mismatch, go, ge, comp, len(q) + len(r) + 1, C.byref(sc))
    assert n >= 0
    return comp.raw[:n], sc.value


def test_aln_ratio_reference_kat(kat):
    L = _lib.load()
    ref, read = kat["min_match"]["ref"].encode(), kat["min_match"]["read"].encode()
    e1 = po.error_rate(b"I" * len(ref), nomin=False)
    e2 = po.error_rate(b"I" * len(read), nomin=False)
    go = L.ioc_host_gap_open(e1 + e2)
    assert go == 5
    comp, _ = align(ref, read, go=go)     # the test aligns (ref, read) in that order
    ratio = L.ioc_host_aln_ratio(comp, len(comp), e1 + e2, len(read), kat["aln_ratio"]["k"])
    assert abs(ratio - kat["aln_ratio"]["expected_double_eq"]) < 1e-15


def test_gap_open_table():
    L = _lib.load()
    assert [L.ioc_host_gap_open(e) for e in (0.0, 0.01, 0.0101, 0.04, 0.041, 0.1, 0.11, 0.5)] == [5, 5, 4, 4, 3, 3, 2, 2]


def test_semi_global_properties():
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(30):
        core = bytes(acgt[rng.integers(0, 4, int(rng.integers(30, 120)))])
        left = bytes(acgt[rng.integers(0, 4, int(rng.integers(0, 40)))])
        right = bytes(acgt[rng.integers(0, 4, int(rng.integers(0, 40)))])
        # a read contained in a longer reference aligns end-gap free with a full-length match run
        comp, score = align(core, left + core + right)
        assert score == 2 * len(core)
        assert comp.count(b"|") == len(core)
        assert len(comp) == len(left) + len(core) + len(right)
        # identical sequences
        comp, score = align(core, core)
        assert comp == b"|" * len(core) and score == 2 * len(core)
        # overlap (suffix of a = prefix of b)
        comp, score = align(left + core, core + right)
        assert score >= 2 * len(core) - 1 and b"|" * min(20, len(core)) in comp
    comp, score = align(b"", b"ACGT")
    assert comp == b"    " and score == 0


def test_aln_ratio_counts_windows_like_the_reference():
    L = _lib.load()
    comp = b"|" * 30 + b" " * 5 + b"|" * 10
    k, e, slen = 13, 0.1, 40
    limit = np.floor((1.0 - e) * k)
    exp = sum(1 for i in range(len(comp) - k) if comp[i:i + k].count(b"|") >= limit) / slen
    assert L.ioc_host_aln_ratio(comp, len(comp), e, slen, k) == exp
