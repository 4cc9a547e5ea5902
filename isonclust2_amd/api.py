"""Python mirror of the host interface of the path (names follow the reference's C++ seam).

Everything here calls the C ABI of libisonclust2_hip.so; numpy arrays are only the carriers of
the flat SoA the ABI takes.  No compute happens in Python and nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BatchView, ClusterStats, IocError, LeftView, Params, Timings

MODE = {"sahlin": 0, "fast": 1, "furious": 2, "none": 3}


def default_params(k=11, w=15, mode="fast"):
    """CmdArgs defaults (src/args.h:9-37) restricted to what the path reads."""
    return Params(k=k, w=w, min_shared=5, mode=MODE[mode], min_fraction=0.8, mapped_threshold=0.65,
                  min_prob_no_hits=0.1, aligned_threshold=0.2)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def host_gap_limits(k, w, min_prob_no_hits=0.1, table=_lib.TABLE_PATH):
    L = _lib.load()
    g = np.zeros(225, np.int32)
    p = np.zeros(225, np.float64)
    rc = L.ioc_host_gap_limits(table.encode(), k, w, min_prob_no_hits, _p(g, C.c_int32), _p(p, C.c_double))
    if rc != 0:
        raise IocError(rc, f"no table rows for k={k}, w={w}")
    return g.reshape(15, 15), p.reshape(15, 15)


def host_err_cell(e):
    return int(_lib.load().ioc_host_err_cell(float(e)))


def host_min_total(hpc_len, thr=0.65):
    return int(_lib.load().ioc_host_min_total(int(hpc_len), float(thr)))


class Context:
    """One context per GPU (ioc_ctx)."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.ioc_ctx_create(device, C.byref(h))
        if rc != 0:
            raise IocError(rc, "ioc_ctx_create failed (no MI355X visible?)")
        self.h = h
        self.device = int(device)   # (the HIP device is per THREAD: helpers that allocate through torch pass this explicitly)
        self._keep = []

    @property
    def serial(self):
        """Generation of the context's queries, kept by the LIBRARY (ioc_queries_generation): it changes with every call
        that replaces them, whichever method made it."""
        return int(self.L.ioc_queries_generation(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.ioc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise IocError(rc, self.L.ioc_last_error(self.h).decode(errors="replace"))
        return rc

    # ---- low level -------------------------------------------------------------------------
    def set_params(self, params: Params, table=_lib.TABLE_PATH):
        g, _ = host_gap_limits(params.k, params.w, params.min_prob_no_hits, table)
        g = np.ascontiguousarray(g.reshape(-1))
        self._chk(self.L.ioc_set_params(self.h, C.byref(params), _p(g, C.c_int32)))
        self.params = params

    def queries_upload(self, off_fwd, off_rev, min_val, min_pos, hpc_len, err_cell, min_total):
        off_fwd = np.ascontiguousarray(off_fwd, np.int64)
        off_rev = np.ascontiguousarray(off_rev, np.int64)
        min_val = np.ascontiguousarray(min_val, np.uint32)
        min_pos = np.ascontiguousarray(min_pos, np.uint32)
        hpc_len = np.ascontiguousarray(hpc_len, np.uint32)
        err_cell = np.ascontiguousarray(err_cell, np.uint8)
        min_total = np.ascontiguousarray(min_total, np.uint32)
        n = len(off_fwd) - 1
        self._chk(self.L.ioc_queries_upload(self.h, n, _p(off_fwd, C.c_int64), _p(off_rev, C.c_int64),
                                            _p(min_val, C.c_uint32), _p(min_pos, C.c_uint32), len(min_val),
                                            _p(hpc_len, C.c_uint32), _p(err_cell, C.c_uint8),
                                            _p(min_total, C.c_uint32)))
        self.n = n

    def left_load(self, n_clusters, cls_err_cell, keys, offs, postings):
        if n_clusters == 0:
            self._chk(self.L.ioc_left_load(self.h, 0, None, 0, None, None, None))
            return
        cls_err_cell = np.ascontiguousarray(cls_err_cell, np.uint8)
        keys = np.ascontiguousarray(keys, np.uint32)
        offs = np.ascontiguousarray(offs, np.int64)
        postings = np.ascontiguousarray(postings, np.uint32)
        self._chk(self.L.ioc_left_load(self.h, n_clusters, _p(cls_err_cell, C.c_uint8), len(keys),
                                       _p(keys, C.c_uint32), _p(offs, C.c_int64), _p(postings, C.c_uint32)))

    def index_update(self, cls, old_min, new_min, new_err_cell=0):
        """UpdateMinDB (src/minimizer.cpp:124-160) for left cluster `cls` on the device."""
        old_min = np.ascontiguousarray(old_min, np.uint32)
        new_min = np.ascontiguousarray(new_min, np.uint32)
        self._chk(self.L.ioc_index_update(self.h, int(cls), _p(old_min, C.c_uint32), len(old_min),
                                          _p(new_min, C.c_uint32), len(new_min), int(new_err_cell)))

    def left_export(self):
        """The left MinDB as it stands on the device: (keys, offs, postings), empty lists included."""
        nk, npost = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.ioc_left_export(self.h, C.byref(nk), C.byref(npost), None, None, None))
        keys = np.zeros(nk.value, np.uint32)
        offs = np.zeros(nk.value + 1, np.int64)
        post = np.zeros(max(1, npost.value), np.uint32)
        self._chk(self.L.ioc_left_export(self.h, C.byref(nk), C.byref(npost), _p(keys, C.c_uint32),
                                         _p(offs, C.c_int64), _p(post, C.c_uint32)))
        return keys, offs, post[:npost.value]

    def index_build(self):
        self._chk(self.L.ioc_index_build(self.h))

    def score(self):
        self._chk(self.L.ioc_score(self.h))

    def resolve(self):
        it = C.c_int32(0)
        self._chk(self.L.ioc_resolve(self.h, C.byref(it)))
        return it.value

    def decisions(self):
        n = self.n
        t, s, f = np.zeros(n, np.int32), np.zeros(n, np.int8), np.zeros(n, np.uint8)
        self._chk(self.L.ioc_get_decisions(self.h, _p(t, C.c_int32), _p(s, C.c_int8), _p(f, C.c_uint8)))
        return t, s, f

    def force_decision(self, q, target, strand=1):
        self._chk(self.L.ioc_force_decision(self.h, q, target, strand))

    def query_candidates(self, q, cap):
        t, s = np.zeros(cap, np.int32), np.zeros(cap, np.int8)
        sz, fi, tm = (np.zeros(cap, np.uint32) for _ in range(3))
        n = self._chk(self.L.ioc_query_candidates(self.h, q, cap, _p(t, C.c_int32), _p(s, C.c_int8),
                                                  _p(sz, C.c_uint32), _p(fi, C.c_uint32), _p(tm, C.c_uint32)))
        return t[:n], s[:n], sz[:n], fi[:n], tm[:n]

    def set_shard(self, world, rank, fn):
        """ioc_set_shard: fast-mode score + resolve of the queries j with j % world == rank only; fn(d_ptr, count, kind,
        stream) is the in-place all-reduce over device memory (kind: _lib.XCHG_*) and returns 0.  world <= 1 or fn None
        switches it off."""
        if fn is None or world <= 1:
            self._shard_cb = None
            self._chk(self.L.ioc_set_shard(self.h, 1, 0, _lib.EXCHANGE_FN(), None))
            return

        def tramp(user, buf, count, kind, stream):
            try:
                return int(fn(buf, count, kind, stream) or 0)
            except Exception:      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._shard_cb = _lib.EXCHANGE_FN(tramp)      # kept alive for as long as the context may call it
        self._chk(self.L.ioc_set_shard(self.h, world, rank, self._shard_cb, None))

    @property
    def shard_exchanges(self):
        return int(self.L.ioc_shard_exchanges(self.h))

    @property
    def shard_aligned_pairs(self):
        """pairs THIS rank aligned in the sharded alignment rounds since set_shard (sahlin / furious)"""
        return int(self.L.ioc_shard_aligned_pairs(self.h))

    def scored_candidates(self, q, cap=1 << 16):
        """(key, size) the scoring kernels wrote for query q: key = target << 1 | strand."""
        key, size = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        m = self._chk(self.L.ioc_scored_candidates(self.h, q, cap, _p(key, C.c_uint32), _p(size, C.c_uint32)))
        assert m <= cap
        return key[:m], size[:m]

    def index_export(self):
        nk, npost = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.ioc_index_export(self.h, C.byref(nk), C.byref(npost), None, None, None))
        keys = np.zeros(max(nk.value, 1), np.uint32)
        offs = np.zeros(nk.value + 1, np.int64)
        post = np.zeros(max(npost.value, 1), np.uint32)
        self._chk(self.L.ioc_index_export(self.h, C.byref(nk), C.byref(npost), _p(keys, C.c_uint32),
                                          _p(offs, C.c_int64), _p(post, C.c_uint32)))
        return keys[:nk.value], offs, post[:npost.value]

    def resident_set_sequences(self, raw_seq, raw_off, raw_err):
        """Raw sequences of the resident queries: lets cluster_resident run sahlin mode."""
        raw_seq = raw_seq if isinstance(raw_seq, bytes) else np.asarray(raw_seq, np.uint8).tobytes()
        raw_off = np.ascontiguousarray(raw_off, np.int64)
        raw_err = np.ascontiguousarray(raw_err, np.float64)
        self._chk(self.L.ioc_resident_set_sequences(self.h, raw_seq, _p(raw_off, C.c_int64), _p(raw_err, C.c_double)))

    # ---- alignment fallback on the GPU ---------------------------------------------------------
    def align_set_pool(self, seqs):
        """Upload the raw sequences (list of bytes) the pairs of align_pairs index into."""
        offs = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum([len(x) for x in seqs], out=offs[1:])
        blob = b"".join(seqs)
        self._chk(self.L.ioc_align_set_pool(self.h, len(seqs), blob, _p(offs, C.c_int64)))

    def align_set_verdict_threshold(self, thr):
        """ioc_align_set_verdict_threshold: > 0 lets tracebacks stop once ratio >= thr is decided (windows / ratio become bounds)."""
        self._chk(self.L.ioc_align_set_verdict_threshold(self.h, float(thr)))

    def align_pairs(self, pairs, k, match=2, mismatch=-2, gap_extend=1):
        """ParasailAlign + getAlnRatio (src/cluster.cpp:408-459) for (query, ref, ref_revcomp, e[, hint]) tuples:
        returns (score, qualifying windows, ratio) arrays."""
        n = len(pairs)
        arr = (_lib.AlnPair * max(n, 1))()
        for i, pr in enumerate(pairs):        # (a fifth element: the similarity hint, ioc_aln_pair::reserved)
            qi, ri, rc, e = pr[:4]
            arr[i].query, arr[i].ref, arr[i].ref_revcomp, arr[i].e = int(qi), int(ri), int(bool(rc)), float(e)
            arr[i].reserved = int(pr[4]) if len(pr) > 4 else 0
        score, win, ratio = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.float64)
        self._chk(self.L.ioc_align_pairs(self.h, n, arr, k, match, mismatch, gap_extend, _p(score, C.c_int32),
                                         _p(win, C.c_int64), _p(ratio, C.c_double)))
        return score, win, ratio

    # ---- sort-stage feeders --------------------------------------------------------------------
    def qual_scores(self, offs, qual, k):
        """CalcQualScore / CalcErrorRate per read (src/qualscore.cpp:14-37, 107-154)."""
        offs = np.ascontiguousarray(offs, np.int64)
        qual = np.ascontiguousarray(qual, np.uint8)
        n = len(offs) - 1
        score, err = np.zeros(n, np.float64), np.zeros(n, np.float64)
        self._chk(self.L.ioc_qual_scores(self.h, n, _p(offs, C.c_int64), _p(qual, C.c_uint8), k,
                                         _p(score, C.c_double), _p(err, C.c_double)))
        return score, err

    def extract_minimizers(self, offs, seq, qual, k, w):
        """HomopolymerCompress + KmerEncodeSeq + GetKmerMinimizers on both strands
        (src/qualscore.cpp:39-105); the minimizers stay on the device."""
        offs = np.ascontiguousarray(offs, np.int64)
        seq = np.ascontiguousarray(seq, np.uint8)
        qual = np.ascontiguousarray(qual, np.uint8)
        n = len(offs) - 1
        hpc_len, hpc_err = np.zeros(n, np.uint32), np.zeros(n, np.float64)
        off_fwd, off_rev = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        status = np.zeros(n, np.int32)
        self._chk(self.L.ioc_extract_minimizers(self.h, n, _p(offs, C.c_int64), _p(seq, C.c_uint8),
                                                _p(qual, C.c_uint8), k, w, _p(hpc_len, C.c_uint32),
                                                _p(hpc_err, C.c_double), _p(off_fwd, C.c_int64),
                                                _p(off_rev, C.c_int64), _p(status, C.c_int32)))
        return dict(hpc_len=hpc_len, hpc_err=hpc_err, off_fwd=off_fwd, off_rev=off_rev, status=status)

    def extracted_download(self, total):
        mn, ps = np.zeros(max(total, 1), np.uint32), np.zeros(max(total, 1), np.uint32)
        self._chk(self.L.ioc_extracted_download(self.h, _p(mn, C.c_uint32), _p(ps, C.c_uint32), len(mn)))
        return mn[:total], ps[:total]

    def queries_from_extracted(self, keep, err_cell, min_total):
        keep = np.ascontiguousarray(keep, np.uint8)
        err_cell = np.ascontiguousarray(err_cell, np.uint8)
        min_total = np.ascontiguousarray(min_total, np.uint32)
        self._chk(self.L.ioc_queries_from_extracted(self.h, _p(keep, C.c_uint8), _p(err_cell, C.c_uint8),
                                                    _p(min_total, C.c_uint32)))
        self.n = len(keep)

    def gather_records_device(self, entries, d_min_ptr, d_pos_ptr, cap):
        """ioc_gather_records_device: the minimizer lists of `entries` of the current queries, gathered on the device
        into the caller's device buffers (addresses as ints, capacity in words).  Returns (words, off_fwd, off_rev)."""
        entries = np.ascontiguousarray(entries, np.int32)
        n = len(entries)
        of, orv = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        w = self.L.ioc_gather_records_device(self.h, n, _p(entries, C.c_int32), C.c_void_p(int(d_min_ptr)), C.c_void_p(int(d_pos_ptr)),
                                             int(cap), _p(of, C.c_int64), _p(orv, C.c_int64))
        self._chk(w)
        return int(w), of, orv

    def timings(self):
        t = Timings()
        self._chk(self.L.ioc_get_timings(self.h, C.byref(t)))
        return t.as_dict()

    def count_reference_postings(self):
        h = C.c_int64(0)
        self._chk(self.L.ioc_count_reference_postings(self.h, C.byref(h)))
        return h.value

    def synchronize(self):
        self._chk(self.L.ioc_synchronize(self.h))

    # ---- ClusterSortedReads on one sorted batch (src/cluster.cpp:67-322) -------------------------
    def cluster_batch(self, params: Params, batch: dict, table=_lib.TABLE_PATH):
        """batch: dict with off_fwd, off_rev, min_val, min_pos, raw_len, hpc_len, score, raw_err,
        hpc_err, state, min_qual (the fields of ioc_batch_view).  Returns (cls, strand, stats)."""
        return self.cluster_merge(params, None, batch, table)

    def _make_view(self, batch: dict):
        """ioc_batch_view over the arrays of `batch` (see _merge_call); returns (view, n, objects that must outlive the call)."""
        on_dev = bool(batch.get("minimizers_on_device"))   # min_val / min_pos: device addresses (ints), total = words
        arrs = {
            "off_fwd": np.ascontiguousarray(batch["off_fwd"], np.int64),
            "off_rev": np.ascontiguousarray(batch["off_rev"], np.int64),
            "min_val": None if on_dev else np.ascontiguousarray(batch["min_val"], np.uint32),
            "min_pos": None if on_dev else np.ascontiguousarray(batch["min_pos"], np.uint32),
            "raw_len": np.ascontiguousarray(batch["raw_len"], np.uint32),
            "hpc_len": np.ascontiguousarray(batch["hpc_len"], np.uint32),
            "score": np.ascontiguousarray(batch["score"], np.float64),
            "raw_err": np.ascontiguousarray(batch["raw_err"], np.float64),
            "hpc_err": np.ascontiguousarray(batch["hpc_err"], np.float64),
            "state": np.ascontiguousarray(batch["state"], np.uint8),
        }
        n = len(arrs["off_fwd"]) - 1
        nm = None
        if batch.get("n_members") is not None:
            nm = np.ascontiguousarray(batch["n_members"], np.int32)
        rseq = roff = None
        if batch.get("raw_seq") is not None:   # sahlin / furious: sequences for the host aligner
            rseq = batch["raw_seq"] if isinstance(batch["raw_seq"], bytes) else np.asarray(batch["raw_seq"], np.uint8).tobytes()
            roff = np.ascontiguousarray(batch["raw_off"], np.int64)
        isc = None
        if batch.get("is_cluster") is not None:
            isc = np.ascontiguousarray(batch["is_cluster"], np.uint8)
        if on_dev:
            mvp = C.cast(C.c_void_p(int(batch["min_val"])), C.POINTER(C.c_uint32))
            mpp = C.cast(C.c_void_p(int(batch["min_pos"])), C.POINTER(C.c_uint32))
            total = int(batch["total"])
        else:
            mvp, mpp, total = _p(arrs["min_val"], C.c_uint32), _p(arrs["min_pos"], C.c_uint32), len(arrs["min_val"])
        v = BatchView(n=n, off_fwd=_p(arrs["off_fwd"], C.c_int64), off_rev=_p(arrs["off_rev"], C.c_int64),
                      min_val=mvp, min_pos=mpp,
                      total=total, raw_len=_p(arrs["raw_len"], C.c_uint32),
                      hpc_len=_p(arrs["hpc_len"], C.c_uint32), score=_p(arrs["score"], C.c_double),
                      raw_err=_p(arrs["raw_err"], C.c_double), hpc_err=_p(arrs["hpc_err"], C.c_double),
                      state=_p(arrs["state"], C.c_uint8), min_qual=float(batch.get("min_qual", 7.0)),
                      raw_seq=rseq, raw_off=_p(roff, C.c_int64) if roff is not None else None,
                      n_members=_p(nm, C.c_int32) if nm is not None else None,
                      depth=int(batch.get("depth", -1)), min_cls_size=int(batch.get("min_cls_size", 3)),
                      is_cluster=_p(isc, C.c_uint8) if isc is not None else None, minimizers_on_device=1 if on_dev else 0)
        return v, n, (arrs, nm, rseq, roff, isc)

    def _merge_call(self, params: Params, left, batch: dict, table, cons=None):
        """ClusterSortedReads(left, right).  left: None (initial clustering) or dict with cls_hpc_err,
        keys, offs, postings (the left clusters' representative error rates + MinDB as CSR).
        batch: the right batch (ioc_batch_view fields; for a clustered right batch one record per
        right cluster = its representative, plus n_members / depth / min_cls_size)."""
        v, n, _alive = self._make_view(batch)
        lv = None
        if left is not None and left.get("resident"):
            # the left state already on the device (left_load + index_update) is used as it is
            le = np.ascontiguousarray(left["cls_hpc_err"], np.float64)
            lv = LeftView(n_clusters=len(le), cls_hpc_err=_p(le, C.c_double), n_keys=-1, keys=None, offs=None,
                          postings=None, rep_seq=None, rep_off=None, cls_raw_err=None)
        elif left is not None:
            le = np.ascontiguousarray(left["cls_hpc_err"], np.float64)
            lk = np.ascontiguousarray(left["keys"], np.uint32)
            lo = np.ascontiguousarray(left["offs"], np.int64)
            lp = np.ascontiguousarray(left["postings"], np.uint32)
            lseq = loff = lerr = None
            if left.get("rep_seq") is not None:
                lseq = left["rep_seq"] if isinstance(left["rep_seq"], bytes) else np.asarray(left["rep_seq"], np.uint8).tobytes()
                loff = np.ascontiguousarray(left["rep_off"], np.int64)
                lerr = np.ascontiguousarray(left["cls_raw_err"], np.float64)
            lv = LeftView(n_clusters=len(le), cls_hpc_err=_p(le, C.c_double), n_keys=len(lk),
                          keys=_p(lk, C.c_uint32), offs=_p(lo, C.c_int64), postings=_p(lp, C.c_uint32),
                          rep_seq=lseq, rep_off=_p(loff, C.c_int64) if loff is not None else None,
                          cls_raw_err=_p(lerr, C.c_double) if lerr is not None else None)
        cls, strand = np.zeros(n, np.int32), np.zeros(n, np.int8)
        st = ClusterStats()
        if cons is None:
            self._chk(self.L.ioc_cluster_merge(self.h, C.byref(params), table.encode(),
                                               C.byref(lv) if lv is not None else None, C.byref(v),
                                               _p(cls, C.c_int32), _p(strand, C.c_int8), C.byref(st)))
        else:
            cargs, ops = cons
            self._chk(self.L.ioc_cluster_consensus(self.h, C.byref(params), table.encode(),
                                                   C.byref(lv) if lv is not None else None, C.byref(v), C.byref(cargs),
                                                   C.byref(ops), _p(cls, C.c_int32), _p(strand, C.c_int8), C.byref(st)))
        self.n = n
        self.params = params
        self._keep = [batch.get("_keepalive")]   # device tensors borrowed by the context (minimizers_on_device)
        return cls, strand, st.as_dict()

    def cluster_merge(self, params: Params, left, batch: dict, table=_lib.TABLE_PATH):
        """ClusterSortedReads(left, right), consensus off (see _merge_call for the views)."""
        return self._merge_call(params, left, batch, table)

    def cluster_consensus(self, params: Params, left, batch: dict, cons_args, ops, table=_lib.TABLE_PATH):
        """ClusterSortedReads with the consensus branch (ioc_cluster_consensus): cons_args = _lib.ConsensusArgs,
        ops = _lib.ConsensusOps (the caller's graph store); batch needs raw_seq / raw_off."""
        return self._merge_call(params, left, batch, table, cons=(cons_args, ops))

    def cluster_resident(self):
        n = self.n
        cls, strand = np.zeros(n, np.int32), np.zeros(n, np.int8)
        st = ClusterStats()
        self._chk(self.L.ioc_cluster_resident(self.h, _p(cls, C.c_int32), _p(strand, C.c_int8), C.byref(st)))
        return cls, strand, st.as_dict()
