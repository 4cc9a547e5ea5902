"""Pins the CPU oracle: (1) against every known-answer vector the reference's own unit tests hold
for the hot path (tests/golden/reference_kat.json <- test/isONclust2_test.cpp), (2) function by
function against oracle/_ref (reference TUs compiled where they lie) when that library was built.
CPU only.
"""
import ctypes as C
import math
import struct

import numpy as np
import pytest

from oracle import pyoracle as po


def ulps(a, b):
    ia = struct.unpack("<q", struct.pack("<d", a))[0]
    ib = struct.unpack("<q", struct.pack("<d", b))[0]
    return abs(ia - ib)


def assert_double_eq(a, b):
    """gtest EXPECT_DOUBLE_EQ = within 4 ULPs."""
    assert ulps(a, b) <= 4, (a, b)


def test_sorting_kat(kat):
    g = kat["sorting"]
    rs = po.ReadSet([r["seq"].encode() for r in g["reads"]], [r["qual"].encode() for r in g["reads"]])
    rs.score_sort(g["k"], g["w"])
    order, score, err = rs.order()
    names = [g["reads"][i]["name"] for i in order]
    assert names == g["expected_order"]
    assert np.all(np.diff(score) <= 0)


def test_minimizer_kat(kat):
    g = kat["minimizer"]
    km = po.kmer_encode(g["seq"].encode(), g["k"])
    assert len(km) == len(g["seq"]) - g["k"]  # the final k-mer is never produced (kmer_index.cpp:12)
    mn, ps, ix = po.minimizers(km, g["k"], g["w"])
    exp = g["expected"]
    assert list(mn) == [e["min"] for e in exp]
    assert list(ps) == [e["pos"] for e in exp]
    assert list(ix) == [e["index"] for e in exp]
    for e in exp:
        assert po.lib().orc_kmer_to_index(e["kmer"].encode(), g["k"]) == e["min"]


def test_hpc_kat(kat):
    g = kat["hpc"]
    s, q = po.hpc(g["seq"].encode(), g["qual"].encode())
    assert s.decode() == g["expected_seq"]
    assert q.decode() == g["expected_qual"]


def test_error_rate_kat(kat):
    g = kat["error_rate"]
    r = po.error_rate(g["qual"].encode(), nomin=False)
    assert_double_eq(r, g["expected_double_eq"])
    assert r == float.fromhex("0x1.a36e2eb1c432fp-14") or ulps(r, 0.0001) <= 4


def test_emp_prob_lookup_kat(kat):
    g = kat["emp_prob_lookup"]
    tab, filled = po.pmin_table(g["k"], g["w"])
    assert filled == 225
    assert_double_eq(po.pmin_lookup(tab, g["e1"], g["e2"]), g["expected_double_eq"])


def test_min_match_kat(kat):
    g = kat["min_match"]
    ref, read = g["ref"].encode(), g["read"].encode()
    top, pe, mr = C.c_uint32(0), C.c_double(0), C.c_double(0)
    rc = po.lib().orc_minmatch(ref, g["qual_char"].encode() * len(ref), len(ref), read,
                               g["qual_char"].encode() * len(read), len(read), g["k"], g["w"],
                               po.PMIN_BIN.encode(), g["min_prob_no_hits"], C.byref(top), C.byref(pe),
                               C.byref(mr))
    assert rc == 0
    assert top.value == g["expected_top_size"]
    assert_double_eq(pe.value, g["expected_p_error_double_eq"])
    assert_double_eq(mr.value, g["expected_mapped_ratio_double_eq"])


def test_kmer_transform_kat(kat):
    k = kat["kmer_transform"]["k"]
    L = po.lib()
    kmers = []
    for i in range(4 ** k):
        b = C.create_string_buffer(k + 1)
        L.orc_index_to_kmer(i, k, b)
        kmers.append(b.raw[:k])
    assert kmers == sorted(kmers)
    assert [L.orc_kmer_to_index(s, k) for s in kmers] == list(range(4 ** k))


def test_gap_limit_matches_pow_predicate():
    """The integer gap limit is exactly the reference predicate pow(pError, n) >= p0
    (cluster.cpp:333-347) for every cell of the (11,16) table and every n up to 4096."""
    tab, _ = po.pmin_table(11, 15)
    for a in range(15):
        for b in range(15):
            lim = po.lib().orc_gap_limit(tab[a, b], 0.1)
            pe = 1.0 - tab[a, b]
            for n in list(range(0, 64)) + [100, 1000, 4096]:
                assert (math.pow(pe, float(n)) >= 0.1) == (n <= lim), (a, b, n, lim)


# ---- cross-checks against the stand-alone reference TUs (oracle/_ref) ---------------------------
needs_ref = pytest.mark.skipif(po.ref() is None, reason="oracle/_ref not built (reference tree absent)")


@needs_ref
def test_ref_kmer_encode_and_revcomp():
    R = po.ref()
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(0, 200))
        k = int(rng.integers(1, 20))
        alpha = b"ACGT" if trial % 5 else b"ACGTN"
        s = bytes(rng.choice(list(alpha), n).astype(np.uint8))
        out = np.zeros(max(n, 1), np.uint32)
        m = R.ref_kmer_encode(s, n, k, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        mine = po.kmer_encode(s, k)
        assert m == len(mine)
        assert np.array_equal(out[:m], mine)
        o = C.create_string_buffer(n + 1)
        rc = R.ref_revcomp(s, n, o)
        if rc == 0:
            assert po.revcomp(s) == o.raw[:n]
        else:
            with pytest.raises(ValueError):
                po.revcomp(s)


@needs_ref
def test_ref_round_and_pmin_all_cells():
    R = po.ref()
    rng = np.random.default_rng(11)
    for x in list(rng.random(2000) * 0.3) + [0.005, 0.015, 0.025, 0.125, 0.145, 0.155, 0.0, 1.0]:
        assert R.ref_round(float(x), 2) == po.lib().orc_round(float(x), 2)
    for k, w in [(11, 15), (13, 20), (10, 12), (15, 19), (30, 32), (31, 40), (12, 200)]:
        h = R.ref_pmin_init(k, w)
        tab, filled = po.pmin_table(k, w)
        assert R.ref_pmin_size(h) == filled
        if filled:
            for e1 in np.linspace(0.0, 0.2, 41):
                for e2 in np.linspace(0.0, 0.2, 41):
                    err = C.c_int(0)
                    r = R.ref_pmin_lookup(h, float(e1), float(e2), C.byref(err))
                    assert err.value == 0
                    assert r == po.pmin_lookup(tab, float(e1), float(e2))
        else:
            err = C.c_int(0)
            R.ref_pmin_lookup(h, 0.05, 0.05, C.byref(err))
            assert err.value == -1
            with pytest.raises(KeyError):
                po.pmin_lookup(tab, 0.05, 0.05)
        R.ref_pmin_free(h)


def test_minimizers_equal_argmin_change_formulation():
    """the reference's sliding rule == 'emit when the leftmost argmin of the window
    changes'.  This is the formulation the HIP kernel uses; pin it against the oracle on
    tie-heavy random input."""
    rng = np.random.default_rng(3)
    for trial in range(400):
        k = int(rng.integers(2, 14))
        w = k + int(rng.integers(0, 8))
        n = int(rng.integers(w - k + 1, 300))
        km = rng.integers(0, int(rng.integers(2, 50)), n).astype(np.uint32)
        mn, ps, ix = po.minimizers(km, k, w)
        W = w - k + 1
        exp = []
        prev = -1
        for j in range(n - W + 1):
            a = j + int(np.argmin(km[j:j + W]))
            if a != prev:
                exp.append((int(km[a]), a))
                prev = a
        assert [(int(a), int(b)) for a, b in zip(mn, ps)] == exp
        assert list(ix) == list(range(len(exp)))
