"""One randomised differential case per call, shared by the developer soaks (tools/fuzz_*.py) and by the bounded slices the
driver runs (tests/test_gpu_fuzz.py).  Every function draws its case from the numpy / random generator it is handed, runs the
product through the C ABI and the CPU oracle on the same input, and returns (ok, tag): `tag` is the reproducer — every
parameter of the case, the seed of its reads included.

What is compared (reference behaviour behind it):
  parity_case      assignments of every entry of a batch, fast / sahlin / furious (src/cluster.cpp:67-322, 355-406, 461-515,
                   622-636), and — fast mode, every fourth case — a two-batch merge (`cluster -l -r`)
  consensus_case   consensus on (src/consensus.cpp:34-126): assignments, the log of graph operations, the event count, the MinDB
  align_batch      the GPU aligner against the host aligner (score + window ratio of getAlnRatio, src/cluster.cpp:408-459)
  verdict_batch    the aligner's verdict mode against its own exact mode
  corridor_batch   long pairs: the forward pass's corridor (probe, certificate, second pass) against every tile and the host aligner
"""
import ctypes as C
import os

import numpy as np

from isonclust2_amd import _lib, api, pipeline, synth
from oracle import pyoracle as po
from tests.helpers import ToyGraphs, oracle_entry_assignments, oracle_sorted_batch


def _with_sequences(rs, view):
    seqs = [rs.read(int(i))[0] for i in view["orig"]]
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    v = dict(view)
    v.update(raw_seq=b"".join(seqs), raw_off=off)
    return v


# ---- clustering parity ---------------------------------------------------------------------------------------------------
def draw_parity(rng, aln_mode=None):
    n = int(rng.integers(1, 260))
    g = int(rng.integers(1, 24))
    ln = int(rng.choice([120, 200, 350, 600, 900, 1500, 2500]))
    qlo = float(rng.choice([7, 9, 11, 14]))
    qhi = qlo + float(rng.choice([4, 8, 12]))
    dup = int(rng.choice([0, 0, 2, 3]))
    jit = float(rng.choice([0.0, 0.0, 0.3]))
    k, w = [(11, 15), (11, 15), (13, 20), (10, 14), (15, 22)][int(rng.integers(0, 5))]
    seed = int(rng.integers(0, 1 << 30))
    if aln_mode:
        n, ln = min(n, 70), min(ln, 350)
    return dict(n=n, g=g, ln=ln, qlo=qlo, qhi=qhi, dup=dup, jit=jit, k=k, w=w, seed=seed, mode=aln_mode or "fast")


def run_parity(ctx, c, merge=False):
    """c: a dict as draw_parity makes it.  Returns (ok, detail)."""
    rs = synth.generate(c["n"], c["g"], c["ln"], c["qlo"], c["qhi"], seed=c["seed"], dup_every=c["dup"], len_jitter=c["jit"])
    k, w, mode = c["k"], c["w"], c["mode"]
    B, view = oracle_sorted_batch(rs, k, w)
    ocl, ost, _ = oracle_entry_assignments(B, view, mode=mode)
    v = _with_sequences(rs, view) if mode != "fast" else view
    cls, strand, st = ctx.cluster_batch(api.default_params(k, w, mode), v)
    if not (np.array_equal(cls, ocl) and np.array_equal(strand, ost)):
        d = np.nonzero((cls != ocl) | (strand != ost))[0]
        return False, f"entries {d[:5].tolist()}: device {cls[d[:5]].tolist()} oracle {ocl[d[:5]].tolist()}"
    if merge and mode == "fast" and c["n"] >= 8:   # a two-batch merge on the same reads
        n = c["n"]
        R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
        R.score_sort(k, w)
        p = po.default_params(k, w)
        cut = n // 2
        obs, cbs = [], []
        for b, (lo, hi) in enumerate(((0, cut - 1), (cut, n - 1))):
            Bo = po.Batch(R, lo, hi, p, batch_nr=b)
            info, off_f, off_r, mn, ps = Bo.minimizer_soa()
            vw = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"], hpc_len=info["hpc_len"],
                      score=info["score"], raw_err=info["raw_err"], hpc_err=info["hpc_err"],
                      state=info["state"].astype(np.uint8), min_qual=p.min_qual)
            sb = pipeline.SortedBatch(view=vw, read_ids=info["orig"].astype(np.int64), batch_nr=b, batch_start=lo, batch_end=hi)
            Bo.cluster(mode="fast")
            obs.append(Bo)
            cbs.append(pipeline.cluster_single(ctx, api.default_params(k, w, "fast"), sb))
        obs[0].cluster(right=obs[1], mode="fast")
        merged = pipeline.cluster_merge(ctx, api.default_params(k, w, "fast"), cbs[0], cbs[1])
        mo, ms = obs[0].assignments(rs.n)
        mc, mst = merged.assignments(rs.n)
        if not (np.array_equal(mc, mo) and np.array_equal(mst, ms)):
            d = np.nonzero((mc != mo) | (mst != ms))[0]
            return False, f"MERGE reads {d[:5].tolist()}: device {mc[d[:5]].tolist()} oracle {mo[d[:5]].tolist()}"
    return True, ""


# ---- consensus -----------------------------------------------------------------------------------------------------------
def draw_consensus(rng, mode="fast"):
    n = int(rng.integers(2, 220))
    g = int(rng.integers(1, 12))
    ln = int(rng.choice([300, 500, 800, 1200]))
    cmax = int(rng.choice([3, 6, 12, 50]))
    cmin = int(rng.choice([2, 3, 5, 20]))
    period = int(rng.choice([5, 25, 500]))
    seed = int(rng.integers(0, 1 << 30))
    dup = int(rng.choice([0, 0, 2]))
    if mode != "fast":
        n, ln = min(n, 120), min(ln, 800)
    return dict(n=n, g=g, ln=ln, cmax=cmax, cmin=cmin, period=period, seed=seed, dup=dup, mode=mode, qlo=11, qhi=22)


def oracle_consensus_run(rs, cons_max, cons_min, period, mode="fast", k=11, w=15, graphs=None, ops_pointer=None):
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    p = po.default_params(k, w)
    p.cons_max_size = cons_max
    B = po.Batch(R, 0, rs.n - 1, p)
    info, off_f, off_r, mn, ps = B.minimizer_soa()
    view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"], hpc_len=info["hpc_len"],
                score=info["score"], raw_err=info["raw_err"], hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8),
                min_qual=p.min_qual, orig=info["orig"])
    g = graphs or ToyGraphs()
    po.lib().orc_set_consensus(ops_pointer or C.cast(C.pointer(g.ops), C.c_void_p), cons_min, period)
    try:
        st = B.cluster(mode=mode)
    finally:
        po.lib().orc_set_consensus(None, 50, 500)
    return B, view, st, g


def run_consensus(ctx, c, speculate=None):
    """speculate: None = the library's default (deferred consensus), False = IOC_CONS_SPECULATE=0 (every event at once)."""
    rs = synth.generate(c["n"], c["g"], c["ln"], c["qlo"], c["qhi"], seed=c["seed"], dup_every=c["dup"])
    mode = c["mode"]
    B, view, ost, og = oracle_consensus_run(rs, c["cmax"], c["cmin"], c["period"], mode=mode)
    acl, ast = B.assignments(rs.n)
    ocl, ostr = acl[view["orig"]], ast[view["orig"]]
    v = _with_sequences(rs, view)
    pg = ToyGraphs()
    cargs = _lib.ConsensusArgs(cons_min_size=c["cmin"], cons_max_size=c["cmax"], cons_period=c["period"], left_depth=-1, left_sizes=None)
    old = os.environ.get("IOC_CONS_SPECULATE")
    if speculate is False:
        os.environ["IOC_CONS_SPECULATE"] = "0"
    try:
        cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, mode), None, v, cargs, pg.ops)
    finally:
        if speculate is False:
            if old is None:
                os.environ.pop("IOC_CONS_SPECULATE", None)
            else:
                os.environ["IOC_CONS_SPECULATE"] = old
    keys, offs, post = ctx.index_export()
    okeys, ooffs, opost = B.index()
    what = []
    if not (np.array_equal(cls, ocl) and np.array_equal(strand, ostr)):
        what.append("assignments")
    if pg.log != og.log:
        what.append("graph-operation log")
    if st["n_cons_invoked"] != ost["cons_invoked"]:
        what.append(f"events {st['n_cons_invoked']} vs {ost['cons_invoked']}")
    if not (np.array_equal(keys, okeys) and np.array_equal(offs, ooffs) and np.array_equal(post, opost)):
        what.append("MinDB")
    return not what, ", ".join(what)


def run_consensus_poa(ctx, c):
    """Consensus mode with REAL partial-order graphs on both sides: the product's engine (ioc_poa.hip) behind
    ioc_cluster_consensus, the oracle's scalar POA (oracle/poa_oracle.cpp) behind the oracle's hook.  Compared: assignments,
    the event count, the MinDB, and every cluster's graph (letters, weighted edges, order) and consensus."""
    from tests.test_gpu_poa import Poa
    rs = synth.generate(c["n"], c["g"], c["ln"], c["qlo"], c["qhi"], seed=c["seed"], dup_every=c["dup"])
    mode = c["mode"]
    o_poa = po.OraclePoa()
    B, view, ost, _ = oracle_consensus_run(rs, c["cmax"], c["cmin"], c["period"], mode=mode, graphs=o_poa, ops_pointer=o_poa.ops_pointer())
    acl, ast = B.assignments(rs.n)
    ocl, ostr = acl[view["orig"]], ast[view["orig"]]
    v = _with_sequences(rs, view)
    p_poa = Poa(ctx)
    what = []
    try:
        cargs = _lib.ConsensusArgs(cons_min_size=c["cmin"], cons_max_size=c["cmax"], cons_period=c["period"], left_depth=-1, left_sizes=None)
        cls, strand, st = ctx.cluster_consensus(api.default_params(11, 15, mode), None, v, cargs, p_poa.ops)
        keys, offs, post = ctx.index_export()
        okeys, ooffs, opost = B.index()
        if not (np.array_equal(cls, ocl) and np.array_equal(strand, ostr)):
            what.append("assignments")
        if st["n_cons_invoked"] != ost["cons_invoked"]:
            what.append(f"events {st['n_cons_invoked']} vs {ost['cons_invoked']}")
        if not (np.array_equal(keys, okeys) and np.array_equal(offs, ooffs) and np.array_equal(post, opost)):
            what.append("MinDB")
        if not what:
            for c_id in range(B.n_clusters()):
                db, dr, def_, det, dew = p_poa.graph(c_id)
                ob, orr, oef, oet, oew = o_poa.graph(c_id)
                if not (db == ob and dr.tolist() == orr.tolist() and
                        sorted(zip(def_.tolist(), det.tolist(), dew.tolist())) == sorted(zip(oef.tolist(), oet.tolist(), oew.tolist()))):
                    what.append(f"graph of cluster {c_id}")
                    break
                if p_poa.consensus(c_id) != o_poa.consensus(c_id):
                    what.append(f"consensus of cluster {c_id}")
                    break
    finally:
        p_poa.close()
        o_poa.close()
    return not what, ", ".join(what)


# ---- aligner -------------------------------------------------------------------------------------------------------------
def align_batch(ctx, rng, lmax=1400, npairs=30):
    """rng: random.Random.  One batch of random pairs through ioc_align_pairs and through the host aligner."""
    from tests.test_gpu_align import _host, _mutate as mut
    L = _lib.load()
    seqs, pairs = [], []
    for t in range(npairs):
        n, m = rng.choice([0, 1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257]) if rng.random() < 0.3 else rng.randint(0, lmax), rng.randint(0, lmax)
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 30))
        kind = rng.random()
        if kind < 0.5:
            q, r = mut(rng, base, rng.choice([0.02, 0.1, 0.25]))[:n], mut(rng, base[rng.randint(0, 20):], 0.1)[:m]
        elif kind < 0.7:
            q, r = bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(m))
        elif kind < 0.85:
            q, r = bytes(rng.choice(b"AC") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(m))
        else:
            q = base[:n]
            r = q[: m] if rng.random() < 0.5 else (b"ACGT" * (m // 4 + 1))[:m]
        if rng.random() < 0.05 and len(q) > 1:
            q = q[: len(q) // 2] + b"N" + q[len(q) // 2 + 1:]      # a letter outside A C G T: the comparing kernel
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, rng.randint(0, 1), rng.choice([0.0, 0.02, 0.05, 0.12, 0.3, 0.95])))
    k = rng.choice([1, 7, 11, 15, 32])
    waves = rng.choice(["1", "2", "4", "8"])
    old = os.environ.get("IOC_ALIGN_WAVES")
    os.environ["IOC_ALIGN_WAVES"] = waves
    try:
        ctx.align_set_pool(seqs)
        score, win, ratio = ctx.align_pairs(pairs, k)
    finally:
        if old is None:
            os.environ.pop("IOC_ALIGN_WAVES", None)
        else:
            os.environ["IOC_ALIGN_WAVES"] = old
    bad = []
    for i, (qi, ri, rc, e) in enumerate(pairs):
        hs, hr = _host(L, seqs[qi], seqs[ri], rc, e, k)
        if score[i] != hs or ratio[i] != hr:
            bad.append((i, len(seqs[qi]), len(seqs[ri]), rc, e, int(score[i]), hs, float(ratio[i]), hr))
    return not bad, f"k={k} waves={waves} pairs {bad[:3]}"


def verdict_batch(ctx, rng, lmax=3000, npairs=60):
    """rng: random.Random.  Verdict mode (early-stopped and parked walks) against the exact mode of the same aligner."""
    from tests.test_gpu_align import _mutate as mut
    seqs, pairs = [], []
    for t in range(npairs):
        n, m = rng.randint(0, lmax), rng.randint(0, lmax)
        if rng.random() < 0.5:
            m = max(0, n + rng.randint(-200, 200))
        base = bytes(rng.choice(b"ACGT") for _ in range(max(n, m) + 30))
        kind = rng.random()
        if kind < 0.45:
            q, r = mut(rng, base, rng.choice([0.02, 0.1, 0.25]))[:n], mut(rng, base[rng.randint(0, 20):], 0.1)[:m]
        elif kind < 0.8:
            q, r = bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(m))
        elif kind < 0.9:
            q, r = bytes(rng.choice(b"AC") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(m))
        else:   # related in one half only: decided late, either way
            q = base[:n]
            r = (base[: m // 2] + bytes(rng.choice(b"ACGT") for _ in range(m)))[:m]
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, rng.randint(0, 1), rng.choice([0.02, 0.05, 0.12, 0.3])))
    k = rng.choice([7, 11, 15])
    ctx.align_set_pool(seqs)
    ctx.align_set_verdict_threshold(0.0)
    bad = []
    try:
        s0, w0, r0 = ctx.align_pairs(pairs, k)
        for thr in (rng.choice([0.05, 0.2, 0.5]), rng.choice([0.1, 0.35, 0.9])):
            ctx.align_set_verdict_threshold(thr)
            s1, w1, r1 = ctx.align_pairs(pairs, k)
            if not (np.array_equal(s0, s1) and np.array_equal(r0 >= thr, r1 >= thr) and bool(np.all(w1 <= w0))):
                x = np.nonzero((s0 != s1) | ((r0 >= thr) != (r1 >= thr)) | (w1 > w0))[0][:3]
                bad.append((thr, [(int(i), len(seqs[2 * i]), len(seqs[2 * i + 1]), int(w0[i]), int(w1[i])) for i in x]))
    finally:
        ctx.align_set_verdict_threshold(0.0)
    return not bad, f"k={k} {bad[:2]}"


def corridor_batch(ctx, rng, lmin=5200, lmax=11000, npairs=14, host_checks=2):
    """rng: random.Random.  Long pairs (the corridor of the aligner's forward pass only exists from ~5 kb on): related ones at several
    divergences, with a long insertion / deletion (the path leaves the main diagonal), a fragment against the whole (the path is
    far off the diagonal: the probe or the certificate must send the pair through every tile), half related, unrelated, low
    complexity, unequal lengths.  The default build (corridor, probe launch, certificate, second pass for the refuted) against
    IOC_ALIGN_CORRIDOR=0 (every tile) in exact mode and in verdict mode; `host_checks` of the pairs also against the host aligner."""
    from tests.test_gpu_align import _host, _mutate as mut
    L = _lib.load()
    seqs, pairs = [], []
    for t in range(npairs):
        n = rng.randint(lmin, lmax)
        base = bytes(rng.choice(b"ACGT") for _ in range(n + 200))
        kind = rng.random()
        if kind < 0.35:
            q, r = mut(rng, base, rng.choice([0.03, 0.08, 0.15]))[:n], mut(rng, base[rng.randint(0, 60):], rng.choice([0.03, 0.1]))
        elif kind < 0.5:      # a long deletion or insertion in the middle
            a, d = rng.randint(n // 4, n // 2), rng.choice([300, 1200, 2500, 4000])
            q, r = mut(rng, base, 0.05), mut(rng, base[:a] + base[a + d:], 0.05)
            if rng.random() < 0.5:
                q, r = r, q
        elif kind < 0.62:     # a fragment against the whole
            a = rng.randint(n // 4, n // 2)
            q, r = mut(rng, base[a:], 0.05), mut(rng, base, 0.05)
            if rng.random() < 0.5:
                q, r = r, q
        elif kind < 0.72:     # related in the first half only
            q, r = mut(rng, base, 0.05), mut(rng, base[: n // 2], 0.05) + bytes(rng.choice(b"ACGT") for _ in range(n // 2))
        elif kind < 0.9:
            q, r = bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(rng.randint(lmin, lmax)))
        else:
            q, r = bytes(rng.choice(b"AC") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(n + rng.randint(-300, 300)))
        seqs += [q, r]
        pairs.append((2 * t, 2 * t + 1, 0, rng.choice([0.02, 0.05, 0.12])))
    k = rng.choice([11, 15])
    thr = rng.choice([0.1, 0.4])
    ctx.align_set_pool(seqs)
    old = os.environ.get("IOC_ALIGN_CORRIDOR")
    res = {}
    try:
        # (the default build twice: the corridor's widths come from the error-rate model, whose line is fitted to the pairs the
        # context has aligned before — the second pass plans with what the first one taught it, whatever that is worth here,
        # where the error rates handed in say nothing about the pairs)
        for frac in ("0", None, "again"):
            if frac is None or frac == "again":
                os.environ.pop("IOC_ALIGN_CORRIDOR", None) if old is None else os.environ.__setitem__("IOC_ALIGN_CORRIDOR", old)
            else:
                os.environ["IOC_ALIGN_CORRIDOR"] = frac
            ctx.align_set_verdict_threshold(0.0)
            exact = ctx.align_pairs(pairs, k)
            ctx.align_set_verdict_threshold(thr)
            verdict = ctx.align_pairs(pairs, k)
            res[frac] = (exact, verdict)
    finally:
        ctx.align_set_verdict_threshold(0.0)
        if old is None:
            os.environ.pop("IOC_ALIGN_CORRIDOR", None)
        else:
            os.environ["IOC_ALIGN_CORRIDOR"] = old
    bad = []
    (e0, v0), (e1, v1) = res["0"], res[None]
    for i in range(npairs):
        (e2, v2) = res["again"]
        if e0[0][i] != e2[0][i] or e0[1][i] != e2[1][i] or e0[2][i] != e2[2][i] or v0[0][i] != v2[0][i] or (v0[2][i] >= thr) != (v2[2][i] >= thr):
            bad.append(("learned", i, len(seqs[2 * i]), len(seqs[2 * i + 1]), int(e0[0][i]), int(e2[0][i]), int(e0[1][i]), int(e2[1][i])))
        if e0[0][i] != e1[0][i] or e0[1][i] != e1[1][i] or e0[2][i] != e1[2][i]:
            bad.append(("exact", i, len(seqs[2 * i]), len(seqs[2 * i + 1]), int(e0[0][i]), int(e1[0][i]), int(e0[1][i]), int(e1[1][i])))
        if v0[0][i] != v1[0][i] or (v0[2][i] >= thr) != (v1[2][i] >= thr) or (e0[2][i] >= thr) != (v1[2][i] >= thr):
            bad.append(("verdict", i, len(seqs[2 * i]), len(seqs[2 * i + 1]), int(v0[0][i]), int(v1[0][i]), float(v0[2][i]), float(v1[2][i])))
    for i in rng.sample(range(npairs), min(host_checks, npairs)):
        hs, hr = _host(L, seqs[2 * i], seqs[2 * i + 1], 0, pairs[i][3], k)
        if e1[0][i] != hs or e1[2][i] != hr:
            bad.append(("host", i, len(seqs[2 * i]), len(seqs[2 * i + 1]), int(e1[0][i]), hs, float(e1[2][i]), hr))
    return not bad, f"k={k} thr={thr} {bad[:3]}"
