"""Shared test plumbing: oracle-side preparation of a sorted batch and comparison helpers.
(The oracle is test infrastructure; the product never imports it.)"""
import numpy as np

from isonclust2_amd import synth
from oracle import pyoracle as po


def oracle_sorted_batch(rs, k=11, w=15, params=None):
    """FillQualScores -> SortByQualScores -> PrepareSortedBatch (one batch holding every read) on the
    oracle; returns (oracle Batch, dict in the layout of ioc_batch_view)."""
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    p = params or po.default_params(k, w)
    B = po.Batch(R, 0, rs.n - 1, p)
    info, off_f, off_r, mn, ps = B.minimizer_soa()
    view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"],
                hpc_len=info["hpc_len"], score=info["score"], raw_err=info["raw_err"],
                hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8), min_qual=p.min_qual,
                orig=info["orig"])
    return B, view


def oracle_entry_assignments(B, view, mode="fast"):
    """Run the oracle's ClusterSortedReads; returns (cls, strand) per batch entry + stats."""
    st = B.cluster(mode=mode)
    n = len(view["orig"])
    acl, ast = B.assignments(int(view["orig"].max()) + 1 if n else 0)
    return acl[view["orig"]], ast[view["orig"]], st


def fnv1a(cls, strand):
    h = 0xcbf29ce484222325
    for c, s in zip(cls.tolist(), strand.tolist()):
        for b in (c & 0xFFFFFFFF).to_bytes(4, "little") + (s & 0xFF).to_bytes(1, "little"):
            h ^= b
            h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
