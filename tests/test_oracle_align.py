"""The oracle's own semi-global aligner (oracle.cpp `sg_trace`, the call of src/cluster.cpp:408-423, 498-503):
pinned by the reference's only alignment known answer (AlnRatioTest, test/isONclust2_test.cpp:137-181) and
cross-checked, comparison string by comparison string, against the product's host aligner — two statements
written independently of each other (parasail itself is absent from the reference tree: tie-breaking unpinned)."""
import ctypes as C
import random

import numpy as np

from isonclust2_amd import _lib, synth
from oracle import pyoracle as po
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch


def _product_host(q, r, go, ge=1):
    L = _lib.load()
    comp = C.create_string_buffer(len(q) + len(r) + 2)
    sc = C.c_int32(0)
    n = L.ioc_host_align(q, len(q), r, len(r), 2, -2, go, ge, comp, len(q) + len(r) + 2, C.byref(sc))
    assert n >= 0
    return sc.value, comp.raw[:n]


def test_oracle_aligner_reproduces_aln_ratio_test(kat):
    ref, read = kat["min_match"]["ref"].encode(), kat["min_match"]["read"].encode()
    e = po.error_rate(b"I" * len(ref), nomin=False) + po.error_rate(b"I" * len(read), nomin=False)
    assert po.lib().orc_gap_open(e) == 5
    _, comp, _ = po.align(ref, read, e, kat["aln_ratio"]["k"])     # the test aligns (ref, read) in that order
    ratio = po.lib().orc_aln_ratio(comp, len(comp), e, len(read), kat["aln_ratio"]["k"])
    assert abs(ratio - kat["aln_ratio"]["expected_double_eq"]) < 1e-15


def _mutate(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            out.append(rng.choice(b"ACGT"))
        elif x < 2 * rate / 3:
            continue
        elif x < rate:
            out.append(ch)
            out.append(rng.choice(b"ACGT"))
        else:
            out.append(ch)
    return bytes(out)


def test_oracle_aligner_equals_product_host_aligner():
    """600 pairs: related, unrelated, low-complexity (many equal-score paths), with other letters, empty / one
    base; every gap-open class.  Score and the whole comparison string are compared."""
    rng = random.Random(5)
    n_checked = 0
    for t in range(600):
        n, m = rng.randint(0, 300), rng.randint(0, 300)
        kind = t % 4
        alpha = b"ACGT" if kind != 2 else b"AC"
        base = bytes(rng.choice(alpha) for _ in range(max(n, m) + 30))
        q = _mutate(rng, base, 0.15)[:n]
        r = (_mutate(rng, base[rng.randint(0, 12):], 0.15) if kind != 1 else bytes(rng.choice(b"ACGT") for _ in range(m)))[:m]
        if kind == 3 and q:
            q = q[: len(q) // 2] + b"N" + q[len(q) // 2 + 1:]
        go = rng.choice([2, 3, 4, 5])
        comp = C.create_string_buffer(len(q) + len(r) + 2)
        sc = C.c_int32()
        ln = po.lib().orc_align(q, len(q), r, len(r), 2, -2, go, 1, comp, len(q) + len(r) + 2, C.byref(sc))
        hs, hc = _product_host(q, r, go)
        assert sc.value == hs, (t, n, m, go)
        assert comp.raw[:ln] == hc, (t, n, m, go)
        n_checked += 1
    assert n_checked == 600


def test_sahlin_with_builtin_aligner_equals_sahlin_through_the_hook():
    """The oracle in sahlin mode with its own aligner against the same run with the product's host aligner behind
    orc_set_aligner: same control flow, same assignments, same number of reads reaching the fallback."""
    L = _lib.load()
    CB = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int)
    hook = CB(lambda read, nread, rep, nrep, go, ge, out, cap:
              L.ioc_host_align(read, nread, rep, nrep, 2, -2, go, ge, C.cast(out, C.c_char_p), cap, None))
    for cfg, seed in (("tiny", 7), ("config1", 1)):
        rs = synth.generate_config(cfg, seed=seed)
        B1, v1 = oracle_sorted_batch(rs)
        c1, s1, st1 = oracle_entry_assignments(B1, v1, mode="sahlin")
        B2, v2 = oracle_sorted_batch(rs)
        po.lib().orc_set_aligner(C.cast(hook, C.c_void_p))
        try:
            c2, s2, st2 = oracle_entry_assignments(B2, v2, mode="sahlin")
        finally:
            po.lib().orc_set_aligner(None)
        assert np.array_equal(c1, c2) and np.array_equal(s1, s2)
        assert st1["aln_invoked"] == st2["aln_invoked"] and st1["aln_invoked"] > 0
