"""oracle/sg_striped.cpp (the 16-bit SSE2 striped score pass bench.py times as the SIMD lower bound of the CPU baseline's
alignment cost) against the oracle's scalar semi-global aligner: the same score on random pairs — related, unrelated, every
gap-open class of src/cluster.cpp:425-441 —, and a saturated pair is reported as such."""
import ctypes as C
import random

from oracle import pyoracle as po


def _mut(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            out.append(rng.choice(b"ACGT"))
        elif x < 2 * rate / 3:
            continue
        elif x < rate:
            out += bytes([ch, rng.choice(b"ACGT")])
        else:
            out.append(ch)
    return bytes(out)


def test_striped_score_equals_the_scalar_aligner():
    L = po.lib()
    rng = random.Random(11)
    n_checked = 0
    for t in range(250):
        n, m = rng.randint(1, 500), rng.randint(1, 500)
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 20))
        q = _mut(rng, base, rng.choice([0.02, 0.1, 0.3]))[:n] if rng.random() < 0.7 else bytes(rng.choice(b"ACGT") for _ in range(n))
        r = _mut(rng, base[rng.randint(0, 10):], 0.1)[:m]
        if not q or not r:
            continue
        go = rng.choice([2, 3, 4, 5])
        comp = C.create_string_buffer(len(q) + len(r) + 2)
        sc, sat = C.c_int32(), C.c_int32()
        L.orc_align(q, len(q), r, len(r), 2, -2, go, 1, comp, len(comp), C.byref(sc))
        assert L.orc_sg_striped16(q, len(q), r, len(r), 2, -2, go, 1, C.byref(sat)) == sc.value and sat.value == 0, (t, len(q), len(r), go)
        n_checked += 1
    assert n_checked > 200


def test_saturation_is_reported():
    L = po.lib()
    s = b"ACGT" * 5000           # 20 000 matches x 2 = 40 000 > 32 767
    sat = C.c_int32()
    assert L.orc_sg_striped16(s, len(s), s, len(s), 2, -2, 3, 1, C.byref(sat)) == -2 ** 31 and sat.value == 1
