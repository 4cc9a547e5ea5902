"""GPU batched aligner (ioc_align_pairs) against the host aligner (ioc_host_align + ioc_host_aln_ratio):
score and getAlnRatio must be identical — same recurrence, tie-breaks and end-cell choice; the device
replaces the traceback matrix by checkpoints + a tiled traceback ("trace", default) or carries the window
statistics forward ("carry") (src/cluster.cpp:408-459)."""
import ctypes as C
import random

import numpy as np
import pytest

from isonclust2_amd import _lib, api

pytestmark = pytest.mark.gpu

_COMP = {65: 84, 67: 71, 71: 67, 84: 65}


def _host(L, q, r, rc, e, k):
    if rc:
        r = bytes(_COMP.get(ch, ch) for ch in reversed(r))
    cap = len(q) + len(r) + 2
    comp = C.create_string_buffer(cap)
    sc = C.c_int32()
    n = L.ioc_host_align(q, len(q), r, len(r), 2, -2, L.ioc_host_gap_open(e), 1, comp, cap, C.byref(sc))
    assert n >= 0
    return sc.value, L.ioc_host_aln_ratio(comp, n, e, len(q), k)


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


def _check(ctx, seqs, pairs, k):
    L = _lib.load()
    ctx.align_set_pool(seqs)
    score, win, ratio = ctx.align_pairs(pairs, k)
    for i, (qi, ri, rc, e) in enumerate(pairs):
        hs, hr = _host(L, seqs[qi], seqs[ri], rc, e, k)
        assert score[i] == hs, (i, len(seqs[qi]), len(seqs[ri]), rc, e, k)
        assert ratio[i] == hr, (i, len(seqs[qi]), len(seqs[ri]), rc, e, k, win[i])


@pytest.fixture(scope="module")
def ctx():
    return api.Context(0)


def test_reference_aln_ratio_vector(ctx, kat):
    """AlnRatioTest (test/isONclust2_test.cpp:137-181): the reference's only alignment known answer."""
    from oracle import pyoracle as po
    ref, read = kat["min_match"]["ref"].encode(), kat["min_match"]["read"].encode()
    e = po.error_rate(b"I" * len(ref), nomin=False) + po.error_rate(b"I" * len(read), nomin=False)
    ctx.align_set_pool([ref, read])
    _, win, _ = ctx.align_pairs([(0, 1, 0, e)], kat["aln_ratio"]["k"])   # the test aligns (ref, read)
    assert abs(win[0] / len(read) - kat["aln_ratio"]["expected_double_eq"]) < 1e-15


@pytest.mark.parametrize("waves", ["1", "2", "8"])
def test_small_random_pairs(ctx, monkeypatch, waves):
    """Lengths 0..200 incl. empty / one base, every gap-open class, limits <= 0 and > k, k = 1..32."""
    monkeypatch.setenv("IOC_ALIGN_WAVES", waves)
    rng = random.Random(11 + int(waves))
    seqs, pairs = [], []
    for t in range(120):
        n, m = rng.randint(0, 200), rng.randint(0, 200)
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 20))
        q = _mutate(rng, base, 0.15)[:n]
        r = _mutate(rng, base[rng.randint(0, 12):], 0.15)[:m]
        if t % 9 == 0:
            q = bytes(rng.choice(b"AC") for _ in range(n))
        if t % 11 == 0:
            r = q
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, t % 2, rng.choice([0.0, 0.02, 0.05, 0.12, 0.3, 0.95, 1.3])))
    for k in (1, 5, 11, 13, 32):
        _check(ctx, seqs, pairs, k)


@pytest.mark.parametrize("waves", ["1", "8"])
def test_multi_strip_pairs(ctx, monkeypatch, waves):
    """References wider than one strip of the workgroup (512 columns per wave): the edge column goes
    through the global scratch; unequal lengths, contained and overlapping pairs."""
    monkeypatch.setenv("IOC_ALIGN_WAVES", waves)
    rng = random.Random(5)
    base = bytes(rng.choice(b"ACGT") for _ in range(12000))
    seqs = [
        _mutate(rng, base[:9000], 0.08), _mutate(rng, base[:9100], 0.10),        # same transcript
        _mutate(rng, base[2000:3000], 0.05), _mutate(rng, base, 0.12),           # short read inside a long one
        _mutate(rng, base[6000:], 0.1), _mutate(rng, base[:7000], 0.1),          # suffix / prefix overlap
        bytes(rng.choice(b"ACGT") for _ in range(4100)), _mutate(rng, base[:4097], 0.02),  # unrelated
    ]
    pairs = [(0, 1, 0, 0.18), (1, 0, 0, 0.18), (2, 3, 0, 0.17), (3, 2, 0, 0.17), (4, 5, 0, 0.2), (6, 7, 0, 0.05),
             (0, 1, 1, 0.18), (7, 0, 0, 0.1)]
    _check(ctx, seqs, pairs, 11)


def test_automatic_width_and_many_pairs(ctx):
    rng = random.Random(3)
    base = [bytes(rng.choice(b"ACGT") for _ in range(1500)) for _ in range(6)]
    seqs = [_mutate(rng, base[i % 6], 0.1) for i in range(60)]
    pairs = [(i, (i + 6) % 60, i % 2, 0.2) for i in range(60)] + [(i, (i + 1) % 60, 0, 0.2) for i in range(60)]
    _check(ctx, seqs, pairs, 11)


@pytest.mark.parametrize("waves", ["1", "2", "4", "8"])
def test_tile_boundaries(ctx, monkeypatch, waves):
    """Lengths around the checkpoint pitch (256) and the strip width (1024): row bands that end exactly on a
    tile, bands without rows, last column on / next to a lane edge, last row on / next to a tile edge."""
    monkeypatch.setenv("IOC_ALIGN_WAVES", waves)
    rng = random.Random(17)
    base = bytes(rng.choice(b"ACGT") for _ in range(2200))
    # (1536 / 1537, 512 / 513: the last strip switches to 8 columns per lane when no more than 512 columns are left)
    lens = [255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1040, 1535, 1536, 1537, 1544, 2047, 2049]
    seqs = [_mutate(rng, base, 0.1)[:ln] for ln in lens] + [base[:1024], base[:1024]]
    pairs = []
    for a in range(len(lens)):
        for b in (a, (a + 5) % len(lens), (a + 8) % len(lens)):
            pairs.append((a, b, (a + b) % 2, rng.choice([0.05, 0.2])))
    pairs.append((len(lens), len(lens) + 1, 0, 0.01))  # identical sequences: one long diagonal
    _check(ctx, seqs, pairs, 11)


@pytest.mark.parametrize("arena", ["lean", "fat"])
def test_checkpoint_arena_slices(ctx, monkeypatch, arena):
    """A 1 MB checkpoint budget forces several launches (slices) over one batch — with the coarse checkpoints of version 2
    (the default: 512-pitch rows and columns of a couple, two pairs to a word) and with version 1's fine ones."""
    monkeypatch.setenv("IOC_ALIGN_CK_BUDGET_MB", "1")
    monkeypatch.setenv("IOC_ALIGN_ARENA", arena)
    rng = random.Random(23)
    base = bytes(rng.choice(b"ACGT") for _ in range(3000))
    seqs = [_mutate(rng, base, 0.1) for _ in range(12)]
    pairs = [(i, (i + 1) % 12, i % 2, 0.2) for i in range(12)]
    _check(ctx, seqs, pairs, 11)
    tm = ctx.timings()
    assert tm["align_version"] == (2 if arena == "lean" else 1) and tm["align_slices"] > 1


def test_lean_arena_is_an_eighth_of_the_fat_one(ctx, monkeypatch):
    """Version 2 keeps (Hq, F*) / (Hq, E*) of every 512th row / column as 16-bit halves of one word per couple; version 1 kept
    every 128th as 32-bit pairs per pair: 4.4 MB instead of 35 MB per 16.7 kb pair.  Same results from both."""
    rng = random.Random(31)
    base = bytes(rng.choice(b"ACGT") for _ in range(6000))
    seqs = [_mutate(rng, base, 0.1) for _ in range(10)]
    pairs = [(i, (i + 1) % 10, i % 2, 0.15) for i in range(10)]
    res, arena = {}, {}
    for mode in ("lean", "fat"):
        monkeypatch.setenv("IOC_ALIGN_ARENA", mode)
        ctx.align_set_pool(seqs)
        res[mode] = ctx.align_pairs(pairs, 11)
        arena[mode] = ctx.timings()["align_arena_bytes"]
    assert np.array_equal(res["lean"][0], res["fat"][0]) and np.array_equal(res["lean"][1], res["fat"][1])
    assert 0 < arena["lean"] * 6 < arena["fat"], arena


def test_couples_of_unequal_pairs_odd_counts_and_other_letters(ctx):
    """Version 2 packs two pairs into every register: pairs of very different sizes in one couple (the short one's bands and
    strips run out first), an odd number of pairs (the last couple has one), empty and tiny sequences, and pairs with letters
    other than A C G T (those take version 1 inside the same call)."""
    rng = random.Random(37)
    lens = [5000, 700, 2300, 1025, 1023, 513, 512, 4097, 33, 1, 0, 1500, 2600]
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(n)) for n in lens]
    seqs += [_mutate(rng, seqs[0], 0.1), _mutate(rng, seqs[2], 0.08), seqs[3][:500] + b"N" + seqs[3][501:], _mutate(rng, seqs[7], 0.12)]
    n = len(seqs)
    pairs = [(i, (i * 5 + 3) % n, i % 2, rng.choice([0.05, 0.2, 0.3])) for i in range(n)] + [(0, 13, 0, 0.2), (14, 2, 1, 0.2), (7, 16, 0, 0.1)]
    assert len(pairs) % 2 == 0
    _check(ctx, seqs, pairs + [(13, 0, 1, 0.2)], 11)     # odd
    _check(ctx, seqs, pairs, 7)


def test_carry_variant(ctx, monkeypatch):
    """The single-pass kernel (window statistics carried forward) against the host aligner too: two
    independent device formulations of the same alignment."""
    monkeypatch.setenv("IOC_ALIGN_VARIANT", "carry")
    rng = random.Random(29)
    base = bytes(rng.choice(b"ACGT") for _ in range(2500))
    seqs = [_mutate(rng, base, 0.12) for _ in range(8)] + [b"", b"A", base[:40]]
    pairs = [(i, (i + 3) % 8, i % 2, rng.choice([0.02, 0.2, 0.95])) for i in range(8)] + [(8, 0, 0, 0.2), (9, 10, 0, 0.2), (10, 1, 1, 0.2)]
    for k in (5, 11, 32):
        _check(ctx, seqs, pairs, k)


def test_full_length_pairs_two_device_formulations_agree(ctx, monkeypatch):
    """BASELINE-size sequences (config 2/3: 16.7 kb reads, the pairs the sahlin fallback really aligns) are too
    slow for the host aligner in a test (~1 s per pair): the two independent device formulations — score-only
    DP + checkpoints + tiled traceback vs window statistics carried through every DP state — must agree on
    score and window count for every pair, same-transcript and unrelated, forward and reverse-complemented;
    a handful of the pairs is also checked against the host aligner."""
    from isonclust2_amd import synth
    rs = synth.generate_config("config2", seed=1)
    rng = random.Random(41)
    ids = rng.sample(range(rs.n), 240)
    seqs = [bytes(rs.read(i)[0]) for i in ids]
    pairs = []
    for t in range(0, 240, 2):
        pairs.append((t, t + 1, t % 4 == 0, 0.05 + 0.2 * rng.random()))          # random partner: mostly unrelated
    tr = {}
    for x, i in enumerate(ids):
        tr.setdefault(int(rs.transcript[i]), []).append(x)                         # ground truth of the generator
    same = [v for v in tr.values() if len(v) >= 2][:60]
    for v in same:   # same transcript: one of the two orientations is the real overlap
        pairs.append((v[0], v[1], 0, 0.2))
        pairs.append((v[0], v[1], 1, 0.2))
    ctx.align_set_pool(seqs)
    s1, w1, _ = ctx.align_pairs(pairs, 11)
    monkeypatch.setenv("IOC_ALIGN_VARIANT", "carry")
    s2, w2, _ = ctx.align_pairs(pairs, 11)
    monkeypatch.delenv("IOC_ALIGN_VARIANT")
    monkeypatch.setenv("IOC_ALIGN_NO_PROFILE", "1")
    s3, w3, _ = ctx.align_pairs(pairs, 11)
    monkeypatch.delenv("IOC_ALIGN_NO_PROFILE")
    monkeypatch.setenv("IOC_ALIGN_PACKED", "1")           # the packed 16-bit forward pass (opt-in)
    s4, w4, _ = ctx.align_pairs(pairs, 11)
    assert ctx.timings()["n_align_refused"] == 0          # every pair stayed inside the 16-bit window
    monkeypatch.delenv("IOC_ALIGN_PACKED")
    monkeypatch.setenv("IOC_ALIGN_FORCE_CROSS", "1")      # every pair split over 2, then 4 workgroups
    s5, w5, _ = ctx.align_pairs(pairs, 11)
    monkeypatch.setenv("IOC_ALIGN_CROSS_GROUPS", "4")
    s6, w6, _ = ctx.align_pairs(pairs[:60], 11)
    assert ctx.timings()["n_align_refused"] == 0
    monkeypatch.delenv("IOC_ALIGN_FORCE_CROSS")
    monkeypatch.delenv("IOC_ALIGN_CROSS_GROUPS")
    assert np.array_equal(s1, s5) and np.array_equal(w1, w5)
    assert np.array_equal(s1[:60], s6) and np.array_equal(w1[:60], w6)
    assert np.array_equal(s1, s2) and np.array_equal(w1, w2)
    assert np.array_equal(s1, s3) and np.array_equal(w1, w3)
    assert np.array_equal(s1, s4) and np.array_equal(w1, w4)
    assert len(set(int(x) for x in w1)) > 20          # not a degenerate comparison
    L = _lib.load()
    for i in (0, len(pairs) - 1, len(pairs) - 2):
        qi, ri, rc, e = pairs[i]
        hs, hr = _host(L, seqs[qi], seqs[ri], rc, e, 11)
        assert s1[i] == hs and w1[i] / len(seqs[qi]) == hr


@pytest.mark.parametrize("waves", ["1", "2", "4"])
def test_packed_forward_pass(ctx, monkeypatch, waves):
    """IOC_ALIGN_PACKED=1: two row bands per wave in the 16-bit halves of the registers, relative scores with a
    moving base, packed row checkpoints.  Same pairs as the tile-boundary, multi-strip and small-pair cases:
    bands without rows, a high half that idles, last column / last row in either half."""
    monkeypatch.setenv("IOC_ALIGN_PACKED", "1")
    monkeypatch.setenv("IOC_ALIGN_WAVES", waves)
    rng = random.Random(23 + int(waves))
    base = bytes(rng.choice(b"ACGT") for _ in range(5200))
    lens = [1, 3, 127, 128, 129, 255, 256, 257, 511, 513, 1023, 1024, 1025, 1040, 2047, 2049, 3100, 4097, 5100]
    seqs = [_mutate(rng, base, 0.1)[:ln] for ln in lens] + [base[:1024], base[:1024], bytes(rng.choice(b"ACGT") for _ in range(3000))]
    pairs = []
    for a in range(len(lens)):
        for b in (a, (a + 5) % len(lens), (a + 8) % len(lens), (a + 13) % len(lens)):
            pairs.append((a, b, (a + b) % 2, rng.choice([0.05, 0.2, 0.95])))
    pairs += [(len(lens), len(lens) + 1, 0, 0.01), (len(lens) + 2, len(lens) - 1, 0, 0.1), (len(lens) - 1, len(lens) + 2, 1, 0.1)]
    for k in (5, 11, 32):
        _check(ctx, seqs, pairs, k)


def test_pairs_split_over_workgroups(ctx, monkeypatch):
    """IOC_ALIGN_FORCE_CROSS=1: every pair the way the tail generation of a big launch is done — split over 2 or 4
    workgroups (8 / 16 row bands), the first band of a workgroup waiting on a flag for the checkpoints of the band
    above.  Lengths around the band boundaries; bands without rows; against the host aligner."""
    monkeypatch.setenv("IOC_ALIGN_FORCE_CROSS", "1")
    rng = random.Random(31)
    base = bytes(rng.choice(b"ACGT") for _ in range(9000))
    lens = [2048, 2049, 2300, 3000, 4096, 4097, 4500, 5000, 6100, 8192, 8200, 9000]
    seqs = [_mutate(rng, base, 0.1)[:ln] for ln in lens]
    pairs = []
    for a in range(len(lens)):
        for b in (a, (a + 3) % len(lens), (a + 7) % len(lens)):
            pairs.append((a, b, (a + b) % 2, rng.choice([0.05, 0.2])))
    _check(ctx, seqs, pairs, 11)
    assert ctx.timings()["n_align_refused"] == 0      # no wait ran out


@pytest.mark.parametrize("force", [True, False])
def test_split_pair_with_longer_reference_than_the_first(ctx, monkeypatch, force):
    """The cross-workgroup flags are indexed with every pair's own strip count: a split pair with FEWER cells but a
    LONGER reference than the slice's first pair (16000 x 8000 = 8 strips, then 3000 x 16000 = 16 strips) must not
    run over its flag slot (round-1 advisor finding).  Both routes that split pairs: forced, and the default route
    for a handful of pairs on an idle chip."""
    if force:
        monkeypatch.setenv("IOC_ALIGN_FORCE_CROSS", "1")
    rng = random.Random(77)
    base = bytes(rng.choice(b"ACGT") for _ in range(16500))
    seqs = [_mutate(rng, base, 0.08)[:16000], _mutate(rng, base, 0.08)[:8000], _mutate(rng, base, 0.08)[:3000],
            _mutate(rng, base, 0.08)[:16000], _mutate(rng, base[5000:], 0.08)[:9000]]
    pairs = [(0, 1, 0, 0.1), (2, 3, 0, 0.1), (2, 0, 0, 0.2), (4, 3, 0, 0.05), (2, 3, 1, 0.1), (1, 0, 0, 0.1)]
    _check(ctx, seqs, pairs, 11)
    assert ctx.timings()["n_align_refused"] == 0


def test_tail_generation_split_in_launch(ctx, monkeypatch):
    """More pairs than one resident generation of workgroups (config 3's 1622 pairs of 16.7 kb): the pairs of the
    tail generation are split over two workgroups each, in the same launch behind the ordinary ones.  Same results
    as with whole tail pairs (IOC_ALIGN_NO_CROSS_TAIL), and no wait ran out."""
    from isonclust2_amd import synth
    rs = synth.generate_config("config2", seed=1)
    rng = random.Random(43)
    ids = rng.sample(range(rs.n), 200)
    seqs = [bytes(rs.read(i)[0]) for i in ids]
    pairs = [(rng.randrange(200), rng.randrange(200), rng.random() < 0.5, 0.05 + 0.2 * rng.random()) for _ in range(1622)]
    ctx.align_set_pool(seqs)
    s1, w1, _ = ctx.align_pairs(pairs, 11)
    assert ctx.timings()["n_align_refused"] == 0
    monkeypatch.setenv("IOC_ALIGN_NO_CROSS_TAIL", "1")
    s2, w2, _ = ctx.align_pairs(pairs, 11)
    assert np.array_equal(s1, s2) and np.array_equal(w1, w2)
    assert len(set(int(x) for x in w1)) > 50


def test_v2_fallbacks(ctx, monkeypatch):
    """The two ways out of version 2: a pair whose scores leave the 16-bit window is flagged by the guard and re-run by version 1
    (forced here by a guard of 40: every pair of this size trips it), and a bounded wait that runs out sends the whole batch
    through version 1 (forced by pre-setting the launch's error word).  Results are the host aligner's either way."""
    rng = random.Random(41)
    base = bytes(rng.choice(b"ACGT") for _ in range(2600))
    seqs = [_mutate(rng, base, 0.1) for _ in range(6)] + [bytes(rng.choice(b"ACGT") for _ in range(1800))]
    pairs = [(i, (i + 1) % 7, i % 2, 0.2) for i in range(7)]
    monkeypatch.setenv("IOC_ALIGN_V2_GUARD", "40")
    t0 = ctx.timings()["n_align_refused"]
    _check(ctx, seqs, pairs, 11)
    t1 = ctx.timings()["n_align_refused"]
    assert t1 - t0 >= 5                      # the guard refused (nearly) every pair; they came back through version 1
    monkeypatch.delenv("IOC_ALIGN_V2_GUARD")
    monkeypatch.setenv("IOC_ALIGN_V2_FAKE_TIMEOUT", "1")
    _check(ctx, seqs, pairs, 11)
    assert ctx.timings()["n_align_refused"] - t1 == len(pairs) and ctx.timings()["align_version"] == 1


def test_verdict_mode_decides_every_comparison_like_the_full_count(ctx):
    """ioc_align_set_verdict_threshold: tracebacks stop once `ratio >= threshold` is decided.  For related pairs (pass early),
    unrelated ones (fail late), pairs near the threshold and degenerate lengths: the comparison comes out as with the exact
    counts at every threshold, the scores are exact, a stopped walk never reports more windows than the full one — and related
    long pairs do stop early (fewer windows than the full count)."""
    rng = random.Random(43)
    base = bytes(rng.choice(b"ACGT") for _ in range(5000))
    other = bytes(rng.choice(b"ACGT") for _ in range(4800))
    seqs = [base, _mutate(rng, base, 0.06), _mutate(rng, base, 0.15), _mutate(rng, base, 0.3), other, _mutate(rng, other, 0.1),
            base[:700] + other[700:3000], base[:40], b"ACGT" * 3, b""]
    n = len(seqs)
    pairs = [(i, j, (i + j) % 2, 0.12) for i in range(n) for j in range(n) if i != j and (i + 2 * j) % 3 != 0]
    ctx.align_set_pool(seqs)
    ctx.align_set_verdict_threshold(0.0)
    s0, w0, r0 = ctx.align_pairs(pairs, 11)
    stopped = 0
    try:
        for thr in (0.2, 0.05, 0.6, 0.95):
            ctx.align_set_verdict_threshold(thr)
            s1, w1, r1 = ctx.align_pairs(pairs, 11)
            assert np.array_equal(s0, s1)
            assert np.array_equal(r0 >= thr, r1 >= thr), thr
            assert np.all(w1 <= w0)
            stopped += int(np.count_nonzero(w1 < w0))
    finally:
        ctx.align_set_verdict_threshold(0.0)
    assert stopped > len(pairs) // 4
    s2, w2, r2 = ctx.align_pairs(pairs, 11)                     # exact again
    assert np.array_equal(w2, w0) and np.array_equal(r2, r0)


@pytest.mark.parametrize("early", ["1", "0"])
def test_second_traceback_launch_takes_turns_when_many_walks_park(ctx, monkeypatch, early):
    """More undecided walks than the second launch has workgroups (one per compute unit): 700 pairs of unrelated 5 kb sequences
    in verdict mode all park after a few blocks and go on with helper waves, two or three per workgroup one after the other;
    among them a few related pairs that are decided in the first launch.  Scores exact, every comparison as with the full count.
    early = 1 (the default): k_fwd2_ends sends the first 256 of the wrong candidates — by their forward score — to a helper
    launch that runs beside the first traceback launch; the rest park as before and go on in the helper launch after it."""
    monkeypatch.setenv("IOC_TRACE2_EARLY", early)
    rng = random.Random(47)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(4500, 5500))) for _ in range(60)]
    seqs += [_mutate(rng, seqs[i], 0.08) for i in range(5)]
    pairs = [(i, (i * 7 + 1 + j) % 60, (i + j) % 2, 0.12) for i in range(60) for j in range(12) if (i * 7 + 1 + j) % 60 != i][:700]
    pairs += [(60 + i, i, 0, 0.12) for i in range(5)] + [(i, 60 + i, 1, 0.12) for i in range(5)]
    ctx.align_set_pool(seqs)
    ctx.align_set_verdict_threshold(0.0)
    s0, w0, r0 = ctx.align_pairs(pairs, 11)
    try:
        ctx.align_set_verdict_threshold(0.2)
        s1, w1, r1 = ctx.align_pairs(pairs, 11)
    finally:
        ctx.align_set_verdict_threshold(0.0)
    assert np.array_equal(s0, s1) and np.array_equal(r0 >= 0.2, r1 >= 0.2) and np.all(w1 <= w0)
    assert int(np.count_nonzero(r0 >= 0.2)) >= 5 and int(np.count_nonzero(r0 < 0.2)) >= 600


def test_similarity_hints_change_the_order_only(ctx):
    """ioc_aln_pair::reserved: pairs hinted far below the call's median are coupled with each other (a couple with an unrelated
    pair gets every tile of its grid, V2Couple).  Right hints, no hints, wrong hints: the same scores, windows and ratios."""
    rng = random.Random(53)
    base = [bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(7000, 9000))) for _ in range(3)]
    seqs, plain, kind = [], [], []
    for t in range(12):
        if t % 3 == 2:
            q, r = base[t % 3], bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(7000, 9000)))
        else:
            q, r = _mutate(rng, base[t % 3], 0.05), _mutate(rng, base[t % 3], 0.08)
        seqs += [q, r]
        plain.append((2 * t, 2 * t + 1, 0, 0.12))
        kind.append(t % 3 == 2)
    ctx.align_set_pool(seqs)
    ref = ctx.align_pairs(plain, 11)
    assert sum(1 for i in range(12) if ref[2][i] < 0.1) == 4               # the unrelated ones
    for hints in ([60 if u else 2500 for u in kind], [2500 if u else 60 for u in kind], [rng.randrange(1, 5000) for _ in kind]):
        got = ctx.align_pairs([p + (h,) for p, h in zip(plain, hints)], 11)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got)), hints


def test_corridor_on_sequences_with_more_than_256_strips(ctx, monkeypatch):
    """A 265 kb pair: 259 strips of 1024 columns (the corridor's strip bounds are 16 bits wide, as V2Item::strip is), 30 bands.
    The corridor's result against every tile."""
    rng = random.Random(59)
    base = bytes(rng.choice(b"ACGT") for _ in range(265000))
    seqs = [_mutate(rng, base, 0.04), _mutate(rng, base, 0.05)]
    ctx.align_set_pool(seqs)
    got = ctx.align_pairs([(0, 1, 0, 0.12)], 11)
    monkeypatch.setenv("IOC_ALIGN_CORRIDOR", "0")
    ref = ctx.align_pairs([(0, 1, 0, 0.12)], 11)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert got[0][0] > 400000 and got[2][0] > 0.9


def test_corridor_with_several_slices(ctx, monkeypatch):
    """Long pairs under a small checkpoint budget: several launches (slices), each with its own probe launch, corridor tiles,
    certificates — and two unrelated pairs among them whose couples get every tile.  Against every tile in one slice."""
    rng = random.Random(61)
    base = bytes(rng.choice(b"ACGT") for _ in range(9000))
    seqs = [_mutate(rng, base, 0.05) for _ in range(10)] + [bytes(rng.choice(b"ACGT") for _ in range(8800)) for _ in range(2)]
    pairs = [(i, (i + 1) % 10, 0, 0.12) for i in range(10)] + [(0, 10, 0, 0.12), (11, 3, 0, 0.12)]
    ctx.align_set_pool(seqs)
    monkeypatch.setenv("IOC_ALIGN_CORRIDOR", "0")
    ref = ctx.align_pairs(pairs, 11)
    monkeypatch.delenv("IOC_ALIGN_CORRIDOR")
    monkeypatch.setenv("IOC_ALIGN_CK_BUDGET_MB", "12")
    got = ctx.align_pairs(pairs, 11)
    assert ctx.timings()["align_slices"] > 1
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
