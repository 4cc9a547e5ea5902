"""Host-side mirror of `isONclust2 cluster` on in-memory batches (src/main.cpp:238-382).

Only bookkeeping lives here (which read sits in which cluster with which strand, the batch
metadata, the representative records a later merge needs); every hot-path computation is a call
into the C ABI (api.Context).  The data layout is the flat SoA of ioc_batch_view, not the
reference's pointer graph.
"""
from dataclasses import dataclass, field

import numpy as np

VIEW_KEYS = ("raw_len", "hpc_len", "score", "raw_err", "hpc_err", "state")


@dataclass
class SortedBatch:
    """What PrepareSortedBatch produces: one record per read of the batch, in sorted order."""
    view: dict               # fields of ioc_batch_view
    read_ids: np.ndarray     # global id of each entry's read
    batch_nr: int = 0
    batch_start: int = 0
    batch_end: int = 0
    depth: int = -1


@dataclass
class ClusteredBatch:
    """A batch after ClusterSortedReads: clusters with their representative records + MinDB."""
    rep_view: dict           # one ioc_batch_view record per cluster (its representative)
    member_cls: np.ndarray   # flat membership: cluster id,
    member_read: np.ndarray  # global read id,
    member_strand: np.ndarray  # MatchStrand
    mindb: tuple             # (keys, offs, postings) CSR
    depth: int = 0
    batch_start: int = 0
    batch_end: int = 0
    stats: dict = field(default_factory=dict)
    rep_seq: bytes = None    # raw sequences of the representatives, concatenated (sahlin / furious merges align them)
    rep_off: np.ndarray = None
    rep_entry: np.ndarray = None   # entry of the sorted batch each cluster's representative came from (fresh batches)
    ctx_serial: int = -1           # Context.serial after the clustering call: the context still holds this batch's queries iff equal

    @property
    def n_clusters(self):
        return len(self.rep_view["hpc_len"])

    def assignments(self, n_reads):
        cls = np.full(n_reads, -1, np.int32)
        strand = np.zeros(n_reads, np.int32)
        cls[self.member_read] = self.member_cls
        strand[self.member_read] = self.member_strand
        return cls, strand


def gather_seqs(seq, off, idx):
    """Sequences idx out of a concatenated byte string with offsets -> (bytes, offsets)."""
    if seq is None:
        return None, None
    a = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, np.uint8)
    off = np.asarray(off, np.int64)
    idx = np.asarray(idx, np.int64)
    lens = (off[1:] - off[:-1])[idx]
    o = np.zeros(len(idx) + 1, np.int64)
    o[1:] = np.cumsum(lens)
    sel = np.repeat(off[:-1][idx] - o[:-1], lens) + np.arange(o[-1])
    return a[sel].tobytes(), o


def sort_stage(ctx, rs, k, w, read_id_base=0, batch_nr=0, min_qual=7.0):
    """`isONclust2 sort` on one read set that becomes ONE batch (src/main.cpp:105-199): FillQualScores ->
    SortByQualScores -> PrepareSortedBatch, the two compute steps on the GPU through the C ABI
    (ioc_qual_scores, ioc_extract_minimizers).  Returns (SortedBatch, order): the batch as host arrays in the
    layout of ioc_batch_view (raw sequences included), order[i] = index in `rs` of entry i."""
    score, err = ctx.qual_scores(rs.offs, rs.qual, k)                       # FillQualScores
    order = np.argsort(-score, kind="stable")                               # SortByQualScores (stable, descending)
    lens = np.diff(rs.offs)[order]
    so = np.zeros(rs.n + 1, np.int64)
    so[1:] = np.cumsum(lens)
    idx = np.repeat(rs.offs[:-1][order] - so[:-1], lens) + np.arange(so[-1])
    seq = rs.seq[idx]
    ex = ctx.extract_minimizers(so, seq, rs.qual[idx], k, w)                # PrepareSortedBatch
    mn, ps = ctx.extracted_download(int(ex["off_rev"][-1]))
    # qualscore.cpp:56-73: placeholder entries (state 1) for reads the sort stage gates
    gated = (ex["status"] != 0) | (-10 * np.log10(err[order]) <= min_qual)
    view = dict(off_fwd=ex["off_fwd"], off_rev=ex["off_rev"], min_val=mn, min_pos=ps,
                raw_len=lens.astype(np.uint32), hpc_len=ex["hpc_len"], score=np.where(ex["status"] != 0, -1.0, score[order]),
                raw_err=err[order], hpc_err=ex["hpc_err"], state=gated.astype(np.uint8), min_qual=min_qual,
                raw_seq=seq.tobytes(), raw_off=so)
    sb = SortedBatch(view=view, read_ids=read_id_base + order.astype(np.int64), batch_nr=batch_nr,
                     batch_start=read_id_base, batch_end=read_id_base + rs.n - 1)
    return sb, order


def gather_records(view, idx):
    """Compact copy of the records `idx` of a view (CSR minimizer lists re-packed, fwd lists first)."""
    idx = np.asarray(idx, np.int64)
    of, orv = np.asarray(view["off_fwd"], np.int64), np.asarray(view["off_rev"], np.int64)
    nf, nr = (of[1:] - of[:-1])[idx], (orv[1:] - orv[:-1])[idx]
    n = len(idx)
    off_f = np.zeros(n + 1, np.int64)
    off_f[1:] = np.cumsum(nf)
    off_r = np.zeros(n + 1, np.int64)
    off_r[1:] = np.cumsum(nr)
    off_r += off_f[-1]

    def take(starts, lens):
        tot = int(lens.sum())
        if tot == 0:
            return np.zeros(0, np.int64)
        base = np.repeat(starts - (np.cumsum(lens) - lens), lens)
        return base + np.arange(tot)

    sel = np.concatenate([take(of[:-1][idx], nf), take(orv[:-1][idx], nr)])
    out = dict(off_fwd=off_f, off_rev=off_r, min_val=np.asarray(view["min_val"])[sel],
               min_pos=np.asarray(view["min_pos"])[sel], min_qual=view.get("min_qual", 7.0))
    for k in VIEW_KEYS:
        out[k] = np.asarray(view[k])[idx]
    return out


def concat_records(a, b):
    """Concatenate two record sets (both compact, fwd lists first)."""
    na, nb = len(a["hpc_len"]), len(b["hpc_len"])
    fa, fb = int(a["off_fwd"][-1]), int(b["off_fwd"][-1])
    ta, tb = len(a["min_val"]), len(b["min_val"])
    off_f = np.concatenate([a["off_fwd"][:-1], b["off_fwd"] + fa])
    rev_a = a["off_rev"] - fa          # relative to start of a's rev block
    rev_b = b["off_rev"] - fb
    rbase = fa + fb
    off_r = np.concatenate([rev_a[:-1] + rbase, rev_b + rbase + (ta - fa)])
    mv = np.concatenate([a["min_val"][:fa], b["min_val"][:fb], a["min_val"][fa:], b["min_val"][fb:]])
    mp = np.concatenate([a["min_pos"][:fa], b["min_pos"][:fb], a["min_pos"][fa:], b["min_pos"][fb:]])
    out = dict(off_fwd=off_f, off_rev=off_r, min_val=mv, min_pos=mp, min_qual=a.get("min_qual", 7.0))
    for k in VIEW_KEYS:
        out[k] = np.concatenate([a[k], b[k]])
    assert len(off_f) == na + nb + 1 and len(off_r) == na + nb + 1
    return out


def cluster_single(ctx, params, sb: SortedBatch, timing=None) -> ClusteredBatch:
    """`cluster -l batch.cer`: initial clustering of one sorted batch (src/main.cpp:262-275).
    timing (optional dict): receives "abi_ms", the wall time of the two C-ABI calls that make up the *core* region of
    SURVEY §8(d) — host arrays in, assignments + MinDB in host arrays out — without this function's bookkeeping."""
    import time
    t0 = time.perf_counter()
    cls, strand, st = ctx.cluster_batch(params, sb.view)
    t1 = time.perf_counter()
    keys, offs, post = ctx.index_export()
    if timing is not None:
        t2 = time.perf_counter()
        timing.update(abi_ms=(t2 - t0) * 1e3, cluster_ms=(t1 - t0) * 1e3, export_ms=(t2 - t1) * 1e3)
    ok = cls >= 0
    # the entry that opened cluster c is its representative (cluster.cpp:178-206)
    n_cls = int(st["n_clusters"])
    rep_entry = np.full(n_cls, -1, np.int64)
    first = np.nonzero(ok)[0]
    # entries are in loop order: the first entry assigned to a NEW id is the one that created it
    seen = np.zeros(n_cls, bool)
    for i in first:
        c = cls[i]
        if not seen[c]:
            seen[c] = True
            rep_entry[c] = i
    rep_view = gather_records(sb.view, rep_entry)
    rep_seq, rep_off = gather_seqs(sb.view.get("raw_seq"), sb.view.get("raw_off"), rep_entry)
    return ClusteredBatch(rep_view=rep_view, rep_seq=rep_seq, rep_off=rep_off, rep_entry=rep_entry.astype(np.int32),
                          ctx_serial=getattr(ctx, "serial", -1), member_cls=cls[ok].astype(np.int32),
                          member_read=np.asarray(sb.read_ids)[ok].astype(np.int64),
                          member_strand=strand[ok].astype(np.int32), mindb=(keys, offs, post),
                          depth=sb.depth + 1 if sb.depth < 0 else sb.depth + 1,
                          batch_start=sb.batch_start, batch_end=sb.batch_end, stats=st)


def cluster_merge(ctx, params, left: ClusteredBatch, right: ClusteredBatch, min_cls_size=3) -> ClusteredBatch:
    """`cluster -l L -r R` (src/main.cpp:247-261 + src/cluster.cpp:67-322): every right cluster's
    representative is matched against the left MinDB; members move with their strands flipped on a
    reverse-strand match; unmatched right clusters are appended to the left."""
    if right.depth > 0 and right.batch_start != left.batch_end + 1:
        raise ValueError("Trying to merge non-consecutive batches! Giving up!")          # cluster.cpp:81-85
    if left.depth > 0 and right.depth > left.depth:
        raise ValueError("The left input batch must have higher depth!")                  # cluster.cpp:87-90
    nR = right.n_clusters
    counts = np.bincount(right.member_cls, minlength=nR).astype(np.int32)
    rv = dict(right.rep_view)
    rv.update(n_members=counts, depth=right.depth, min_cls_size=min_cls_size)
    lv = dict(cls_hpc_err=left.rep_view["hpc_err"], keys=left.mindb[0], offs=left.mindb[1],
              postings=left.mindb[2])
    if left.rep_seq is not None and right.rep_seq is not None:   # sahlin / furious: getBestClusterAln aligns representatives
        lv.update(rep_seq=left.rep_seq, rep_off=left.rep_off, cls_raw_err=left.rep_view["raw_err"])
        rv.update(raw_seq=right.rep_seq, raw_off=right.rep_off)
    cls, strand, st = ctx.cluster_merge(params, lv, rv)
    keys, offs, post = ctx.index_export()
    L = left.n_clusters
    kept = cls >= 0
    new = np.nonzero(kept & (cls >= L))[0]
    # new left clusters appear in creation order; a right cluster is "new" iff its id is its own
    created = []
    seen = set()
    for i in new:
        if cls[i] not in seen:
            seen.add(int(cls[i]))
            created.append(i)
    rep_view = left.rep_view
    rep_seq, rep_off = left.rep_seq, left.rep_off
    if created:
        rep_view = concat_records(left.rep_view, gather_records(right.rep_view, np.array(created)))
        if left.rep_seq is not None and right.rep_seq is not None:
            add, ao = gather_seqs(right.rep_seq, right.rep_off, np.array(created))
            rep_seq = left.rep_seq + add
            rep_off = np.concatenate([left.rep_off, ao[1:] + left.rep_off[-1]])
    mcl = cls[right.member_cls]
    mst = strand[right.member_cls].astype(np.int32) * right.member_strand
    keep = mcl >= 0
    return ClusteredBatch(rep_view=rep_view, rep_seq=rep_seq, rep_off=rep_off,
                          member_cls=np.concatenate([left.member_cls, mcl[keep].astype(np.int32)]),
                          member_read=np.concatenate([left.member_read, right.member_read[keep]]),
                          member_strand=np.concatenate([left.member_strand, mst[keep]]),
                          mindb=(keys, offs, post), depth=left.depth + 1, batch_start=left.batch_start,
                          batch_end=right.batch_end, stats=st)


# ---- consensus mode (ConsMaxSize > 0): the same bookkeeping with representatives that change ---------------------------
def slice_sorted(sb: SortedBatch, a, b, batch_nr=0) -> SortedBatch:
    """Entries a..b-1 of a sorted read set as a batch of their own (`sort` cuts the globally sorted reads into batches of
    consecutive entries, src/main.cpp:149-199)."""
    idx = np.arange(a, b)
    v = gather_records(sb.view, idx)
    seq, off = gather_seqs(sb.view.get("raw_seq"), sb.view.get("raw_off"), idx)
    v.update(raw_seq=seq, raw_off=off)
    ids = np.asarray(sb.read_ids)[a:b]
    return SortedBatch(view=v, read_ids=ids, batch_nr=batch_nr, batch_start=sb.batch_start + a, batch_end=sb.batch_start + b - 1)


def _patch_reps(rep_view, rep_seq, rep_off, rep_records):
    """Replace the records of the clusters whose representative became a consensus (ioc_rep_record of the LAST event of
    each cluster: src/consensus.cpp:93-124 rewrites RawSeq, HpcSeq, Mins, RevMins and both error rates in place)."""
    last = {}
    for c, r in rep_records:
        last[c] = r
    if not last:
        return rep_view, rep_seq, rep_off
    cl = sorted(last)
    rs = [last[c] for c in cl]
    nf = np.array([len(r["fwd_min"]) for r in rs], np.int64)
    nr = np.array([len(r["rev_min"]) for r in rs], np.int64)
    off_f = np.zeros(len(rs) + 1, np.int64)
    off_f[1:] = np.cumsum(nf)
    off_r = np.zeros(len(rs) + 1, np.int64)
    off_r[1:] = np.cumsum(nr)
    off_r += off_f[-1]
    cat = lambda key: np.concatenate([r[key] for r in rs]).astype(np.uint32)
    patch = dict(off_fwd=off_f, off_rev=off_r, min_val=np.concatenate([cat("fwd_min"), cat("rev_min")]),
                 min_pos=np.concatenate([cat("fwd_pos"), cat("rev_pos")]), min_qual=rep_view.get("min_qual", 7.0),
                 raw_len=np.array([r["raw_len"] for r in rs], np.uint32), hpc_len=np.array([r["hpc_len"] for r in rs], np.uint32),
                 score=np.array([r["score"] for r in rs], np.float64), raw_err=np.array([r["raw_err"] for r in rs], np.float64),
                 hpc_err=np.array([r["hpc_err"] for r in rs], np.float64), state=np.zeros(len(rs), np.uint8))
    n = len(rep_view["hpc_len"])
    idx = np.arange(n)
    idx[np.array(cl)] = n + np.arange(len(cl))
    both = concat_records({k: (np.asarray(v) if k != "min_qual" else v) for k, v in rep_view.items() if k in patch}, patch)
    out = gather_records(both, idx)
    pseq = b"".join(r["raw_seq"] for r in rs)
    poff = np.zeros(len(rs) + 1, np.int64)
    poff[1:] = np.cumsum([len(r["raw_seq"]) for r in rs])
    seq, off = gather_seqs(rep_seq + pseq, np.concatenate([rep_off, poff[1:] + rep_off[-1]]), idx)
    return out, seq, off


def cluster_consensus_single(ctx, params, sb: SortedBatch, cons, store) -> ClusteredBatch:
    """`cluster -l batch.cer` of a batch sorted with `-c ConsMaxSize > 0` (src/cluster.cpp:263-309): ioc_cluster_consensus
    with the caller's graph store (`store.ops`: ioc_consensus_ops; `store.rep_records` receives every replaced
    representative).  cons = (ConsMinSize, ConsMaxSize, ConsPeriod)."""
    from . import _lib
    n0 = len(store.rep_records)
    cargs = _lib.ConsensusArgs(cons_min_size=cons[0], cons_max_size=cons[1], cons_period=cons[2], left_depth=-1, left_sizes=None)
    cls, strand, st = ctx.cluster_consensus(params, None, sb.view, cargs, store.ops)
    keys, offs, post = ctx.index_export()
    ok = cls >= 0
    n_cls = int(st["n_clusters"])
    ent = np.nonzero(ok)[0]
    uniq, first = np.unique(cls[ent], return_index=True)   # entries are in loop order: the first one of an id created it
    rep_entry = np.full(n_cls, -1, np.int64)
    rep_entry[uniq] = ent[first]
    rep_view = gather_records(sb.view, rep_entry)
    rep_seq, rep_off = gather_seqs(sb.view.get("raw_seq"), sb.view.get("raw_off"), rep_entry)
    rep_view, rep_seq, rep_off = _patch_reps(rep_view, rep_seq, rep_off, store.rep_records[n0:])
    return ClusteredBatch(rep_view=rep_view, rep_seq=rep_seq, rep_off=rep_off, member_cls=cls[ok].astype(np.int32),
                          member_read=np.asarray(sb.read_ids)[ok].astype(np.int64), member_strand=strand[ok].astype(np.int32),
                          mindb=(keys, offs, post), depth=0, batch_start=sb.batch_start, batch_end=sb.batch_end, stats=st)


def cluster_consensus_merge(ctx, params, left: ClusteredBatch, right: ClusteredBatch, cons, store, min_cls_size=3) -> ClusteredBatch:
    """`cluster -l L -r R` with consensus on: leftBatch->Depth != -1, so ConsMinSize is 2 and ConsPeriod is ignored
    (src/cluster.cpp:267-288); a right cluster's graph contributes its sequence count as the weight of the addition
    (src/consensus.cpp:51-53, 76-81).  store: side 0 = the left clusters' graphs, side 1 = the right clusters'."""
    from . import _lib
    import ctypes as C
    if right.depth > 0 and right.batch_start != left.batch_end + 1:
        raise ValueError("Trying to merge non-consecutive batches! Giving up!")          # cluster.cpp:81-85
    if left.depth > 0 and right.depth > left.depth:
        raise ValueError("The left input batch must have higher depth!")                  # cluster.cpp:87-90
    n0 = len(store.rep_records)
    nR = right.n_clusters
    counts = np.bincount(right.member_cls, minlength=nR).astype(np.int32)
    lsizes = (np.bincount(left.member_cls, minlength=left.n_clusters) + 1).astype(np.int32)   # + the representative copy
    rv = dict(right.rep_view)
    rv.update(n_members=counts, depth=right.depth, min_cls_size=min_cls_size, raw_seq=right.rep_seq, raw_off=right.rep_off)
    lv = dict(cls_hpc_err=left.rep_view["hpc_err"], keys=left.mindb[0], offs=left.mindb[1], postings=left.mindb[2],
              rep_seq=left.rep_seq, rep_off=left.rep_off, cls_raw_err=left.rep_view["raw_err"])
    cargs = _lib.ConsensusArgs(cons_min_size=cons[0], cons_max_size=cons[1], cons_period=cons[2], left_depth=left.depth,
                               left_sizes=lsizes.ctypes.data_as(C.POINTER(C.c_int32)))
    cls, strand, st = ctx.cluster_consensus(params, lv, rv, cargs, store.ops)
    keys, offs, post = ctx.index_export()
    L = left.n_clusters
    new = np.nonzero(cls >= L)[0]
    uniq, first = np.unique(cls[new], return_index=True)
    created = new[first]                                   # right clusters that became left clusters, in id order
    assert np.array_equal(uniq, L + np.arange(len(uniq)))
    rep_view, rep_seq, rep_off = left.rep_view, left.rep_seq, left.rep_off
    if len(created):
        rep_view = concat_records(left.rep_view, gather_records(right.rep_view, created))
        add, ao = gather_seqs(right.rep_seq, right.rep_off, created)
        rep_seq = left.rep_seq + add
        rep_off = np.concatenate([left.rep_off, ao[1:] + left.rep_off[-1]])
    rep_view, rep_seq, rep_off = _patch_reps(rep_view, rep_seq, rep_off, store.rep_records[n0:])
    mcl = cls[right.member_cls]
    mst = strand[right.member_cls].astype(np.int32) * right.member_strand
    keep = mcl >= 0
    return ClusteredBatch(rep_view=rep_view, rep_seq=rep_seq, rep_off=rep_off,
                          member_cls=np.concatenate([left.member_cls, mcl[keep].astype(np.int32)]),
                          member_read=np.concatenate([left.member_read, right.member_read[keep]]),
                          member_strand=np.concatenate([left.member_strand, mst[keep]]),
                          mindb=(keys, offs, post), depth=left.depth + 1, batch_start=left.batch_start,
                          batch_end=right.batch_end, stats=st)
