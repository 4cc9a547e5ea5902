"""The POA engine behind the consensus operations (ioc_poa_*).  spoa is absent from the
reference tree, so there is no reference answer to pin: checked here are (a) the GPU sequence-to-graph DP
against a plain-Python restatement of the same recurrence on the exported graph (score; the returned path is a
valid walk through the graph whose own score equals it), (b) that the consensus of noisy copies recovers the
sequence they came from, (c) the whole consensus-mode pipeline with this engine on both sides (oracle control
flow vs product driver)."""
import ctypes as C
import random

import numpy as np
import pytest

from isonclust2_amd import _lib, api

pytestmark = pytest.mark.gpu

SC = dict(m=4, n=-8, g=-8, e=-4, q=-20, c=-1)   # src/main.cpp:285-290
NEG = -10 ** 9


class Poa:
    def __init__(self, ctx):
        self.L = _lib.load()
        self.h = C.c_void_p()
        rc = self.L.ioc_poa_create(ctx.h, SC["m"], SC["n"], SC["g"], SC["e"], SC["q"], SC["c"], C.byref(self.h))
        assert rc == 0
        self.ops = _lib.ConsensusOps()
        self.L.ioc_poa_bind(self.h, C.byref(self.ops))

    def create(self, idx, s, side=0):
        assert self.ops.create(self.ops.user, side, idx, C.cast(C.c_char_p(s), C.POINTER(C.c_char)), len(s)) == 0

    def add(self, idx, s, w=1, side=0):
        assert self.ops.add(self.ops.user, side, idx, C.cast(C.c_char_p(s), C.POINTER(C.c_char)), len(s), w) == 0

    def size(self, idx, side=0):
        return self.ops.size(self.ops.user, side, idx)

    def consensus(self, idx, side=0):
        buf = C.create_string_buffer(1 << 20)
        n = self.ops.consensus(self.ops.user, side, idx, C.cast(buf, C.POINTER(C.c_char)), len(buf))
        assert n >= 0
        return buf.raw[:n]

    def graph(self, idx, side=0):
        nn, ne = C.c_int32(), C.c_int32()
        assert self.L.ioc_poa_graph_export(self.h, side, idx, C.byref(nn), C.byref(ne), None, None, None, None, None) == 0
        bases = C.create_string_buffer(nn.value + 1)
        rank = np.zeros(max(1, nn.value), np.int32)
        ef, et = np.zeros(max(1, ne.value), np.int32), np.zeros(max(1, ne.value), np.int32)
        ew = np.zeros(max(1, ne.value), np.int64)
        p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.L.ioc_poa_graph_export(self.h, side, idx, C.byref(nn), C.byref(ne), bases, p32(rank), p32(ef), p32(et),
                                           ew.ctypes.data_as(C.POINTER(C.c_int64))) == 0
        return bases.raw[:nn.value], rank[:nn.value], ef[:ne.value], et[:ne.value], ew[:ne.value]

    def last_alignment(self):
        sc = C.c_int32()
        n = self.L.ioc_poa_last_alignment(self.h, 0, None, None, C.byref(sc))
        nodes, pos = np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
        p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.L.ioc_poa_last_alignment(self.h, n, p32(nodes), p32(pos), C.byref(sc)) == n
        return nodes[:n], pos[:n], sc.value

    def close(self):
        self.L.ioc_poa_destroy(self.h)


from tests.poa_common import _path_score, _ref_score  # noqa: E402,F401


def _mutate(rng, s, rate):
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


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def test_graph_dp_equals_the_plain_recurrence(ctx):
    rng = random.Random(7)
    poa = Poa(ctx)
    truth = bytes(rng.choice(b"ACGT") for _ in range(180))
    poa.create(0, _mutate(rng, truth, 0.1))
    for t in range(9):
        read = _mutate(rng, truth, 0.15)
        if t == 4:
            read = read[40:120]                     # a fragment: local alignment, prefix/suffix unaligned
        if t == 6:
            read = bytes(rng.choice(b"ACGT") for _ in range(60)) + read[:100]   # unrelated head
        bases, rank, ef, et, ew = poa.graph(0)      # the graph the read is aligned to
        want = _ref_score(bases, rank, ef, et, read)
        poa.add(0, read, w=1 + t % 3)
        nodes, pos, score = poa.last_alignment()
        assert score == want, (t, score, want)
        assert _path_score(bases, ef, et, read, nodes, pos) == score, t
    assert poa.size(0) == 10
    bases, rank, ef, et, ew = poa.graph(0)
    assert sorted(rank.tolist()) == list(range(len(bases)))                      # a permutation
    order = {int(v): i for i, v in enumerate(rank)}
    assert all(order[int(a)] < order[int(b)] for a, b in zip(ef, et))            # topological
    poa.close()


def test_consensus_recovers_the_source_sequence(ctx):
    rng = random.Random(3)
    poa = Poa(ctx)
    truth = bytes(rng.choice(b"ACGT") for _ in range(900))
    reads = [_mutate(rng, truth, 0.12) for _ in range(14)]
    poa.create(5, reads[0])
    d0 = _edit_distance(reads[0], truth)
    for r in reads[1:]:
        poa.add(5, r)
    cons = poa.consensus(5)
    d = _edit_distance(cons, truth)
    assert d <= 0.03 * len(truth) and d < d0 / 3, (d, d0)
    # purge: the graph restarts from one sequence carrying the old count as weight
    ops = poa.ops
    assert ops.purge(ops.user, 0, 5, C.cast(C.c_char_p(cons), C.POINTER(C.c_char)), len(cons), 14) == 0
    assert poa.size(5) == 1 and poa.consensus(5) == cons
    poa.close()


def test_consensus_mode_pipeline_with_the_poa_engine(ctx):
    """ioc_cluster_consensus with this engine against the oracle's consensus branch driving the ORACLE's scalar POA
    (oracle/poa_oracle.cpp): assignments, event counts, the final MinDB and every final graph."""
    from isonclust2_amd import synth
    from oracle import pyoracle as po
    rs = synth.generate(120, 4, 700, 12, 21, seed=2)
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    p = po.default_params(11, 15)
    p.cons_max_size = 10
    B = po.Batch(R, 0, rs.n - 1, p)
    info, off_f, off_r, mn, ps = B.minimizer_soa()
    view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"], hpc_len=info["hpc_len"],
                score=info["score"], raw_err=info["raw_err"], hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8),
                min_qual=p.min_qual, orig=info["orig"])
    o_poa = po.OraclePoa()
    po.lib().orc_set_consensus(o_poa.ops_pointer(), 3, 500)
    try:
        ost = B.cluster(mode="fast")
    finally:
        po.lib().orc_set_consensus(None, 50, 500)
    assert ost["cons_invoked"] > 10
    acl, ast = B.assignments(rs.n)
    ocl, ostr = acl[view["orig"]], ast[view["orig"]]
    seqs = [rs.read(int(i))[0] for i in view["orig"]]
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    v = dict(view)
    v.update(raw_seq=b"".join(seqs), raw_off=off)
    p_poa = Poa(ctx)
    cargs = _lib.ConsensusArgs(cons_min_size=3, cons_max_size=10, cons_period=500, left_depth=-1, left_sizes=None)
    cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, "fast"), None, v, cargs, p_poa.ops)
    assert st["n_cons_invoked"] == ost["cons_invoked"]
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ostr)
    keys, offs, post = ctx.index_export()
    okeys, ooffs, opost = B.index()
    assert np.array_equal(keys, okeys) and np.array_equal(offs, ooffs) and np.array_equal(post, opost)
    for c_id in range(B.n_clusters()):      # the graphs themselves: letters, edges with their weights, order, consensus
        db, dr, def_, det, dew = p_poa.graph(c_id)
        ob, orr, oef, oet, oew = o_poa.graph(c_id)
        assert db == ob and dr.tolist() == orr.tolist(), c_id
        assert sorted(zip(def_.tolist(), det.tolist(), dew.tolist())) == sorted(zip(oef.tolist(), oet.tolist(), oew.tolist())), c_id
        assert p_poa.consensus(c_id) == o_poa.consensus(c_id), c_id
    o_poa.close()
    p_poa.close()


def test_graph_dp_across_tiles(ctx):
    """Reads longer than one tile column (1024 columns) against a graph deeper than one tile row (64 nodes): row
    carries and boundary columns cross tiles, a long deletion spans a tile edge."""
    rng = random.Random(19)
    poa = Poa(ctx)
    truth = bytes(rng.choice(b"ACGT") for _ in range(2150))
    poa.create(0, _mutate(rng, truth, 0.08))
    reads = [_mutate(rng, truth, 0.1), _mutate(rng, truth[:1000] + truth[1060:], 0.05)]   # the second lacks 60 bases at the tile edge
    for t, read in enumerate(reads):
        bases, rank, ef, et, ew = poa.graph(0)
        want = _ref_score(bases, rank, ef, et, read)
        poa.add(0, read)
        nodes, pos, score = poa.last_alignment()
        assert score == want, (t, score, want)
        assert _path_score(bases, ef, et, read, nodes, pos) == score, t
    poa.close()


def _check_adds(poa, idx, reads):
    for t, read in enumerate(reads):
        bases, rank, ef, et, ew = poa.graph(idx)
        want = _ref_score(bases, rank, ef, et, read)
        poa.add(idx, read)
        nodes, pos, score = poa.last_alignment()
        assert score == want, (t, score, want)
        assert _path_score(bases, ef, et, read, nodes, pos) == score, t


def test_graph_dp_predecessors_out_of_the_lds_ring(ctx):
    """Deletions of 20-45 bases put edges into the graph whose source is more than 16 rows above the target inside
    one 64-row tile (read back from memory, past the L1) or in the tile above; later reads use those edges."""
    rng = random.Random(23)
    poa = Poa(ctx)
    truth = bytes(rng.choice(b"ACGT") for _ in range(420))
    poa.create(0, truth)
    cut = lambda s, a, n: s[:a] + s[a + n:]
    reads = [cut(truth, 70, 24), cut(cut(truth, 70, 24), 200, 45), _mutate(rng, cut(truth, 70, 24), 0.05),
             cut(truth, 130, 20), _mutate(rng, cut(cut(truth, 130, 20), 290, 33), 0.04), _mutate(rng, truth, 0.08)]
    _check_adds(poa, 0, reads)
    bases, rank, ef, et, ew = poa.graph(0)
    order = {int(v): i for i, v in enumerate(rank)}
    spans = [order[int(b)] - order[int(a)] for a, b in zip(ef, et)]
    assert max(spans) > 16                                           # the case is in the graph
    poa.close()


def test_graph_dp_with_a_tiny_predecessor_staging_area(ctx, monkeypatch):
    """IOC_POA_PRED_LDS = 5: all but the first predecessor entries of a tile come from memory, not from LDS."""
    monkeypatch.setenv("IOC_POA_PRED_LDS", "5")
    rng = random.Random(29)
    poa = Poa(ctx)
    truth = bytes(rng.choice(b"ACGT") for _ in range(300))
    poa.create(0, _mutate(rng, truth, 0.1))
    _check_adds(poa, 0, [_mutate(rng, truth, 0.15) for _ in range(5)] + [truth[:120] + truth[150:]])
    poa.close()


def _run_consensus(ctx, rs, mode, cons, window=None, speculate=True, monkeypatch=None):
    """ioc_cluster_consensus on the product's own sort stage with the POA engine; returns everything comparable."""
    from isonclust2_amd import pipeline
    sb, _ = pipeline.sort_stage(ctx, rs, 11, 15)
    events = []

    def rep_changed(user, cls, rec):
        r = rec.contents
        events.append((int(cls), int(r.entry), C.string_at(r.raw_seq, r.raw_len), int(r.hpc_len), int(r.n_fwd), int(r.n_rev)))

    monkeypatch.setenv("IOC_CONS_SPECULATE", "1" if speculate else "0")
    monkeypatch.setenv("IOC_CONS_VIEW_CHECK", "1")   # every pass: the patched left view against a rebuilt one (an error if they differ)
    if window:
        monkeypatch.setenv("IOC_CONS_WINDOW", str(window))
    else:
        monkeypatch.delenv("IOC_CONS_WINDOW", raising=False)
    poa = Poa(ctx)
    cb = _lib.CONS_REP_CHANGED(rep_changed)
    poa.ops.rep_changed = cb
    cargs = _lib.ConsensusArgs(cons_min_size=cons[0], cons_max_size=cons[1], cons_period=cons[2], left_depth=-1, left_sizes=None)
    cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, mode), None, sb.view, cargs, poa.ops)
    db = ctx.index_export()
    graphs = {}
    for c in range(int(st["n_clusters"])):
        if poa.size(c) >= 0:
            g = poa.graph(c)
            graphs[c] = (g[0], tuple(g[1].tolist()), tuple(g[2].tolist()), tuple(g[3].tolist()), tuple(g[4].tolist()), poa.size(c))
    poa.close()
    return cls, strand, st, events, db, graphs


@pytest.mark.parametrize("shape,mode,cons,window", [((300, 6, 600), "fast", (3, 12, 500), None), ((300, 6, 600), "fast", (3, 12, 500), 7),
                                                    ((260, 10, 500), "sahlin", (2, 8, 40), None), ((400, 5, 400), "fast", (4, 1000, 500), 64),
                                                    ((220, 3, 900), "sahlin", (3, 6, 500), 16)])
def test_deferred_consensus_equals_immediate(ctx, monkeypatch, shape, mode, cons, window):
    """The driver with deferred consensus requests (ioc_consensus_spec_ops: events of many clusters of a pass batched,
    verification afterwards, rollback on a violation) against the same driver taking every consensus at once, as the
    reference does: assignments, the sequence of representative replacements with their consensus sequences, the final
    MinDB and every cluster's final graph must be the same."""
    from isonclust2_amd import synth
    rs = synth.generate(shape[0], shape[1], shape[2], 11, 21, seed=sum(shape) + len(mode), dup_every=2 if shape[1] > 8 else 0)
    a = _run_consensus(ctx, rs, mode, cons, window, speculate=False, monkeypatch=monkeypatch)
    b = _run_consensus(ctx, rs, mode, cons, window, speculate=True, monkeypatch=monkeypatch)
    assert a[2]["n_cons_invoked"] > 5
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2]["n_cons_invoked"] == b[2]["n_cons_invoked"] and a[2]["n_clusters"] == b[2]["n_clusters"]
    assert a[3] == b[3]                                    # every representative replacement, in order, with its consensus
    for x, y in zip(a[4], b[4]):
        assert np.array_equal(x, y)                        # MinDB
    assert a[5].keys() == b[5].keys()
    for c in a[5]:
        assert a[5][c] == b[5][c], c                       # nodes, order, weighted edges, sequence count of every graph
    assert b[2]["n_cons_restarts"] <= a[2]["n_cons_restarts"]
    # rollbacks forced at every 5th / 2nd re-checked entry (graphs created, extended and purged inside rolled-back
    # stretches): a rollback only repeats work
    for every in ("5", "2"):
        monkeypatch.setenv("IOC_CONS_FORCE_ROLLBACK", every)
        c = _run_consensus(ctx, rs, mode, cons, window, speculate=True, monkeypatch=monkeypatch)
        monkeypatch.delenv("IOC_CONS_FORCE_ROLLBACK")
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and a[3] == c[3]
        for x, y in zip(a[4], c[4]):
            assert np.array_equal(x, y)
        assert a[5] == c[5]


@pytest.mark.gpu
def test_graphs_travel_as_blobs_one_by_one_and_many_at_once(ctx):
    """ioc_poa_graph_save -> ioc_poa_graph_load / ioc_poa_graph_load_many (the blobs of a merge's thousands of graphs are parsed on
    the host's cores): the loaded graphs equal the saved ones — nodes, order, weighted edges, sequence count, consensus —, and a
    batch with one refused blob loads nothing."""
    rng = random.Random(21)
    src = Poa(ctx)
    L = src.L
    blobs = []
    for g in range(6):
        truth = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([120, 300, 700])))
        src.create(g, _mutate(rng, truth, 0.05))
        for _ in range(rng.randint(0, 5)):
            src.add(g, _mutate(rng, truth, 0.1))
        sz = L.ioc_poa_graph_save(src.h, 0, g, None, 0)
        assert sz > 0
        buf = (C.c_uint8 * sz)()
        assert L.ioc_poa_graph_save(src.h, 0, g, buf, sz) == sz
        blobs.append(buf)

    def same(dst, side, idx, g):
        a, b = src.graph(g), dst.graph(idx, side)
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
        assert src.size(g) == dst.size(idx, side) and src.consensus(g) == dst.consensus(idx, side)

    one = Poa(ctx)
    for g, buf in enumerate(blobs):
        assert L.ioc_poa_graph_load(one.h, 1, 10 + g, buf, len(buf)) == 0
        same(one, 1, 10 + g, g)
    many = Poa(ctx)
    ids = (C.c_int32 * 6)(*[5 - g for g in range(6)])
    ptrs = (C.POINTER(C.c_uint8) * 6)(*[C.cast(b, C.POINTER(C.c_uint8)) for b in blobs])
    lens = (C.c_int64 * 6)(*[len(b) for b in blobs])
    assert L.ioc_poa_graph_load_many(many.h, 0, 6, ids, ptrs, lens) == 0
    for g in range(6):
        same(many, 0, 5 - g, g)
    # one blob cut short: refused, and none of the batch is there
    none = Poa(ctx)
    lens_bad = (C.c_int64 * 6)(*[len(b) if g != 3 else len(b) // 2 for g, b in enumerate(blobs)])
    assert L.ioc_poa_graph_load_many(none.h, 0, 6, ids, ptrs, lens_bad) != 0
    assert all(none.size(g) < 0 for g in range(6))
    assert L.ioc_poa_graph_load_many(none.h, 0, 0, None, None, None) == 0
    # a graph of the builds before round 4's edge weights ("IOCPOA1") is refused by name, not grown further
    assert bytes(blobs[0][:8]) == b"IOCPOA2\x00"
    old = (C.c_uint8 * len(blobs[0]))(*bytes(blobs[0]))
    old[6] = ord("1")
    assert L.ioc_poa_graph_load(none.h, 0, 0, old, len(old)) != 0
    assert b"earlier build" in L.ioc_last_error(ctx.h)
    for p in (src, one, many, none):
        p.close()
