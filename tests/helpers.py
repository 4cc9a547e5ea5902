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


from isonclust2_amd.digest import fnv1a  # noqa: E402,F401  (one definition for tests, goldens and bench)


class ToyGraphs:
    """A stand-in for the per-cluster spoa graphs of the consensus (spoa is absent from the reference tree):
    the same five operations, deterministic and cheap — a "graph" is the list of (sequence, weight) it was fed,
    its consensus the heaviest sequence (ties: the latest), truncated by one base per call so that the
    representative really changes.  The SAME store semantics serve the oracle (orc_set_consensus) and the
    product (ioc_cluster_consensus): what the tests pin is everything AROUND the graphs."""

    def __init__(self, right_sizes=None):
        import ctypes as C
        from isonclust2_amd import _lib
        self.g = {0: {}, 1: {}}
        self.calls = 0
        self.rep_events = []
        self.rep_records = []  # (cluster, full ioc_rep_record as a dict of copies): what a later merge needs of a replaced representative
        self.log = []          # (operation, side, idx, sequence length / weight): compared between the two sides
        for i, s in (right_sizes or {}).items():
            self.g[1][i] = [(b"", 1)] * s
        self._C = C

        def create(user, side, idx, seq, n):
            self.g[side][idx] = [(C.string_at(seq, n), 1)]
            self.log.append(("create", side, idx, n))
            return 0

        def size(user, side, idx):
            return len(self.g[side][idx]) if idx in self.g[side] else -1

        def add(user, side, idx, seq, n, weight):
            if idx not in self.g[side]:
                return -1
            self.g[side][idx].append((C.string_at(seq, n), int(weight)))
            self.log.append(("add", side, idx, n, int(weight)))
            return 0

        def consensus(user, side, idx, out, cap):
            self.calls += 1
            items = self.g[side][idx]
            best = max(range(len(items)), key=lambda t: (items[t][1], t))
            s = items[best][0]
            s = s[: max(64, len(s) - (self.calls % 7))]
            if len(s) > cap:
                return -1
            C.memmove(out, s, len(s))
            self.log.append(("consensus", side, idx, len(s)))
            return len(s)

        def purge(user, side, idx, seq, n, weight):
            self.g[side][idx] = [(C.string_at(seq, n), int(weight))]
            self.log.append(("purge", side, idx, n, int(weight)))
            return 0

        def rep_changed(user, cls, rec):
            r = rec.contents
            self.rep_events.append((int(cls), int(r.entry), C.string_at(r.raw_seq, r.raw_len), float(r.raw_err), float(r.hpc_err),
                                    int(r.hpc_len), int(r.n_fwd), int(r.n_rev)))
            import numpy as _np
            take = lambda ptr, n: _np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else _np.zeros(0, _np.uint32)
            self.rep_records.append((int(cls), dict(
                raw_seq=C.string_at(r.raw_seq, r.raw_len), raw_len=int(r.raw_len), raw_err=float(r.raw_err),
                score=float(r.raw_score), hpc_len=int(r.hpc_len), hpc_err=float(r.hpc_err),
                fwd_min=take(r.fwd_min, r.n_fwd), fwd_pos=take(r.fwd_pos, r.n_fwd),
                rev_min=take(r.rev_min, r.n_rev), rev_pos=take(r.rev_pos, r.n_rev), entry=int(r.entry))))

        self._keep = (_lib.CONS_CREATE(create), _lib.CONS_SIZE(size), _lib.CONS_ADD(add), _lib.CONS_CONSENSUS(consensus),
                      _lib.CONS_PURGE(purge), _lib.CONS_REP_CHANGED(rep_changed))
        self.ops = _lib.ConsensusOps(None, *self._keep)

    def reset(self, right_sizes=None):
        self.g = {0: {}, 1: {}}
        self.calls = 0
        self.rep_events = []
        self.rep_records = []
        self.log = []
        for i, s in (right_sizes or {}).items():
            self.g[1][i] = [(b"", 1)] * s
