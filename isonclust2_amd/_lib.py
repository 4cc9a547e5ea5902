"""ctypes loader of libisonclust2_hip.so (the C ABI of include/isonclust2_hip.h).

There is no CPU fallback: if the HIP library is missing, cannot be loaded, or no GPU is visible,
every compute entry point raises.  The library is built in-tree by `__graft_entry__.build()` /
`make -C isonclust2_amd/csrc`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IOC_LIB", os.path.join(HERE, "libisonclust2_hip.so"))  # IOC_LIB: developer builds
TABLE_PATH = os.path.join(HERE, "data", "pmin_shared.bin")


class IocError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"isonclust2_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("min_shared", C.c_int32), ("mode", C.c_int32),
                ("min_fraction", C.c_double), ("mapped_threshold", C.c_double),
                ("min_prob_no_hits", C.c_double), ("aligned_threshold", C.c_double)]


class BatchView(C.Structure):
    _fields_ = [("n", C.c_int32), ("off_fwd", C.POINTER(C.c_int64)), ("off_rev", C.POINTER(C.c_int64)),
                ("min_val", C.POINTER(C.c_uint32)), ("min_pos", C.POINTER(C.c_uint32)),
                ("total", C.c_int64), ("raw_len", C.POINTER(C.c_uint32)),
                ("hpc_len", C.POINTER(C.c_uint32)), ("score", C.POINTER(C.c_double)),
                ("raw_err", C.POINTER(C.c_double)), ("hpc_err", C.POINTER(C.c_double)),
                ("state", C.POINTER(C.c_uint8)), ("min_qual", C.c_double),
                ("raw_seq", C.c_char_p), ("raw_off", C.POINTER(C.c_int64)),
                ("n_members", C.POINTER(C.c_int32)), ("depth", C.c_int32), ("min_cls_size", C.c_int32),
                ("is_cluster", C.POINTER(C.c_uint8)), ("minimizers_on_device", C.c_int32)]


class LeftView(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("cls_hpc_err", C.POINTER(C.c_double)), ("n_keys", C.c_int64),
                ("keys", C.POINTER(C.c_uint32)), ("offs", C.POINTER(C.c_int64)),
                ("postings", C.POINTER(C.c_uint32)), ("rep_seq", C.c_char_p),
                ("rep_off", C.POINTER(C.c_int64)), ("cls_raw_err", C.POINTER(C.c_double))]


class ClusterStats(C.Structure):
    _fields_ = [("n_clusters", C.c_int64), ("n_joined", C.c_int64), ("n_gated", C.c_int64),
                ("n_tie_replays", C.c_int64), ("n_aln_invoked", C.c_int64),
                ("resolve_iters", C.c_int32), ("aln_rounds", C.c_int32), ("n_aln_pairs", C.c_int64),
                ("n_aln_order_dep", C.c_int64), ("n_cons_invoked", C.c_int64), ("n_cons_restarts", C.c_int64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RepRecord(C.Structure):  # ioc_rep_record
    _fields_ = [("raw_seq", C.POINTER(C.c_char)), ("raw_len", C.c_int32), ("raw_qual", C.c_char), ("raw_err", C.c_double),
                ("raw_score", C.c_double), ("hpc_seq", C.POINTER(C.c_char)), ("hpc_len", C.c_int32), ("hpc_err", C.c_double),
                ("fwd_min", C.POINTER(C.c_uint32)), ("fwd_pos", C.POINTER(C.c_uint32)), ("n_fwd", C.c_int32),
                ("rev_min", C.POINTER(C.c_uint32)), ("rev_pos", C.POINTER(C.c_uint32)), ("n_rev", C.c_int32),
                ("entry", C.c_int32)]


CONS_CREATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int)
CONS_SIZE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
CONS_ADD = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int, C.c_uint)
CONS_CONSENSUS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int)
CONS_PURGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int, C.c_uint)
CONS_REP_CHANGED = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(RepRecord))


class ConsensusOps(C.Structure):  # ioc_consensus_ops (the oracle's orc_cons_ops is its first six members)
    _fields_ = [("user", C.c_void_p), ("create", CONS_CREATE), ("size", CONS_SIZE), ("add", CONS_ADD),
                ("consensus", CONS_CONSENSUS), ("purge", CONS_PURGE), ("rep_changed", CONS_REP_CHANGED),
                ("spec", C.c_void_p)]   # ioc_consensus_spec_ops* (NULL: no deferred consensus; ioc_poa_bind sets it)


class ConsensusArgs(C.Structure):  # ioc_consensus_args
    _fields_ = [("cons_min_size", C.c_int32), ("cons_max_size", C.c_int32), ("cons_period", C.c_int32),
                ("left_depth", C.c_int32), ("left_sizes", C.POINTER(C.c_int32))]


class AlnPair(C.Structure):  # ioc_aln_pair
    _fields_ = [("query", C.c_int32), ("ref", C.c_int32), ("ref_revcomp", C.c_int32), ("reserved", C.c_int32),
                ("e", C.c_double)]


class DistMergeTimes(C.Structure):  # ioc_dist_merge_times
    _fields_ = [("ms_exchange_lists", C.c_float), ("ms_merge", C.c_double), ("bytes_lists", C.c_int64), ("bytes_records", C.c_int64),
                ("sharded", C.c_int32), ("exchanges", C.c_int32)]


# ioc_exchange_fn (ioc_set_shard): user, device buffer, element count, kind, hip stream
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
XCHG_MAX_U8, XCHG_MIN_U32, XCHG_SUM_I32 = 0, 1, 2


class Timings(C.Structure):
    _fields_ = [("ms_build", C.c_float), ("ms_score", C.c_float), ("ms_resolve", C.c_float),
                ("resolve_iters", C.c_int32), ("n_queries", C.c_int32), ("n_minimizers", C.c_int64),
                ("n_index_postings", C.c_int64), ("n_candidates", C.c_int64),
                ("n_mapped_evals", C.c_int64), ("postings_traversed", C.c_int64),
                ("ms_align_fwd", C.c_float), ("ms_align_trace", C.c_float), ("n_align_pairs", C.c_int64),
                ("n_align_cells", C.c_int64),
                ("n_align_refused", C.c_int64), ("score_oob", C.c_int32), ("score_oob_probe", C.c_int32),
                ("align_arena_bytes", C.c_int64), ("align_slices", C.c_int32), ("align_version", C.c_int32),
                ("n_align_cells_computed", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/isonclust2_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ioc_ctx_create", "ioc_ctx_destroy", "ioc_ctx_trim", "ioc_ctx_prewarm", "ioc_last_error", "ioc_set_stream", "ioc_synchronize",
    "ioc_set_params", "ioc_queries_upload", "ioc_queries_bind_device", "ioc_left_load",
    "ioc_index_build", "ioc_score", "ioc_resolve", "ioc_get_decisions", "ioc_get_cuts", "ioc_force_decision",
    "ioc_clear_forced", "ioc_query_candidates", "ioc_index_export", "ioc_qual_scores",
    "ioc_extract_minimizers", "ioc_extracted_download", "ioc_extracted_hpc_download", "ioc_queries_from_extracted",
    "ioc_get_timings", "ioc_count_reference_postings", "ioc_host_gap_limits", "ioc_host_err_cell", "ioc_host_min_total",
    "ioc_cluster_batch", "ioc_cluster_merge", "ioc_cluster_resident", "ioc_host_align", "ioc_host_gap_open",
    "ioc_host_aln_ratio", "ioc_align_set_pool", "ioc_align_pairs", "ioc_set_aln_verdicts", "ioc_get_ties", "ioc_resident_set_sequences",
    "ioc_index_update", "ioc_left_export", "ioc_cluster_consensus",
    "ioc_poa_create", "ioc_poa_destroy", "ioc_poa_bind", "ioc_poa_graph_export", "ioc_poa_last_alignment",
    "ioc_poa_graph_save", "ioc_poa_graph_load", "ioc_poa_graph_load_many", "ioc_gather_records_device", "ioc_queries_generation", "ioc_scored_candidates",
    "ioc_dist_unique_id", "ioc_dist_init", "ioc_dist_shutdown", "ioc_dist_info", "ioc_dist_allgather_device",
    "ioc_dist_allgatherv_device", "ioc_dist_allgather_i64", "ioc_dist_allgatherv_host", "ioc_dist_allreduce_max",
    "ioc_dist_barrier", "ioc_dist_merge", "ioc_set_shard", "ioc_shard_exchanges", "ioc_shard_aligned_pairs", "ioc_dist_exchange", "ioc_dist_set_shard", "ioc_align_set_verdict_threshold",
]

_lib = None


def load():
    """Load the HIP library or raise (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IocError(-100, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # torch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both own the
    # GPU.  Importing torch first makes the dynamic loader resolve our DT_NEEDED libamdhip64 to the copy
    # torch already mapped, so both sides share one runtime whatever the initialisation order.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    pi64, pu32, pu8 = C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    pi32, pi8, pd = C.POINTER(C.c_int32), C.POINTER(C.c_int8), C.POINTER(C.c_double)
    L.ioc_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.ioc_ctx_destroy.argtypes = [vp]
    L.ioc_ctx_destroy.restype = None
    L.ioc_last_error.argtypes = [vp]
    L.ioc_last_error.restype = C.c_char_p
    L.ioc_set_stream.argtypes = [vp, vp]
    L.ioc_synchronize.argtypes = [vp]
    L.ioc_set_params.argtypes = [vp, C.POINTER(Params), pi32]
    L.ioc_queries_upload.argtypes = [vp, i32, pi64, pi64, pu32, pu32, i64, pu32, pu8, pu32]
    L.ioc_queries_bind_device.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, vp, vp, pi64, pi64]
    L.ioc_gather_records_device.argtypes = [vp, i32, pi32, vp, vp, i64, pi64, pi64]
    L.ioc_gather_records_device.restype = i64
    L.ioc_scored_candidates.argtypes = [vp, i32, i32, pu32, pu32]
    L.ioc_queries_generation.argtypes = [vp]
    L.ioc_queries_generation.restype = i64
    L.ioc_left_load.argtypes = [vp, i32, pu8, i64, pu32, pi64, pu32]
    L.ioc_index_update.argtypes = [vp, i32, pu32, i64, pu32, i64, C.c_uint8]
    L.ioc_left_export.argtypes = [vp, pi64, pi64, pu32, pi64, pu32]
    L.ioc_index_build.argtypes = [vp]
    L.ioc_score.argtypes = [vp]
    L.ioc_resolve.argtypes = [vp, pi32]
    L.ioc_get_decisions.argtypes = [vp, pi32, pi8, pu8]
    L.ioc_get_cuts.argtypes = [vp, pi32]
    L.ioc_force_decision.argtypes = [vp, i32, i32, i32]
    L.ioc_clear_forced.argtypes = [vp]
    L.ioc_query_candidates.argtypes = [vp, i32, i32, pi32, pi8, pu32, pu32, pu32]
    L.ioc_index_export.argtypes = [vp, pi64, pi64, pu32, pi64, pu32]
    L.ioc_qual_scores.argtypes = [vp, i32, pi64, pu8, i32, pd, pd]
    L.ioc_extract_minimizers.argtypes = [vp, i32, pi64, pu8, pu8, i32, i32, pu32, pd, pi64, pi64, pi32]
    L.ioc_extracted_download.argtypes = [vp, pu32, pu32, i64]
    L.ioc_extracted_hpc_download.argtypes = [vp, C.c_char_p, C.c_char_p, i64]
    L.ioc_queries_from_extracted.argtypes = [vp, pu8, pu8, pu32]
    L.ioc_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.ioc_count_reference_postings.argtypes = [vp, pi64]
    L.ioc_host_gap_limits.argtypes = [C.c_char_p, i32, i32, C.c_double, pi32, pd]
    L.ioc_host_err_cell.argtypes = [C.c_double]
    L.ioc_host_err_cell.restype = C.c_uint8
    L.ioc_host_min_total.argtypes = [C.c_uint32, C.c_double]
    L.ioc_host_min_total.restype = C.c_uint32
    L.ioc_cluster_batch.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(BatchView), pi32, pi8,
                                    C.POINTER(ClusterStats)]
    L.ioc_cluster_merge.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(LeftView), C.POINTER(BatchView),
                                    pi32, pi8, C.POINTER(ClusterStats)]
    L.ioc_cluster_consensus.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(LeftView), C.POINTER(BatchView),
                                        C.POINTER(ConsensusArgs), C.POINTER(ConsensusOps), pi32, pi8, C.POINTER(ClusterStats)]
    L.ioc_poa_create.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.ioc_poa_destroy.argtypes = [vp]
    L.ioc_poa_destroy.restype = None
    L.ioc_poa_bind.argtypes = [vp, C.POINTER(ConsensusOps)]
    L.ioc_poa_bind.restype = None
    L.ioc_poa_graph_export.argtypes = [vp, C.c_int, C.c_int, pi32, pi32, C.c_char_p, pi32, pi32, pi32, pi64]
    L.ioc_poa_last_alignment.argtypes = [vp, i32, pi32, pi32, pi32]
    L.ioc_poa_graph_save.argtypes = [vp, C.c_int, C.c_int, pu8, i64]
    L.ioc_poa_graph_save.restype = C.c_int64
    L.ioc_poa_graph_load.argtypes = [vp, C.c_int, C.c_int, pu8, i64]
    L.ioc_poa_graph_load_many.argtypes = [vp, C.c_int, C.c_int32, C.POINTER(C.c_int32), C.POINTER(pu8), C.POINTER(i64)]
    L.ioc_host_align.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, i32, i32, i32, C.c_char_p, i32, pi32]
    L.ioc_host_gap_open.argtypes = [C.c_double]
    L.ioc_host_aln_ratio.argtypes = [C.c_char_p, i32, C.c_double, C.c_uint32, C.c_uint32]
    L.ioc_host_aln_ratio.restype = C.c_double
    L.ioc_cluster_resident.argtypes = [vp, pi32, pi8, C.POINTER(ClusterStats)]
    L.ioc_resident_set_sequences.argtypes = [vp, C.c_char_p, pi64, pd]
    L.ioc_set_aln_verdicts.argtypes = [vp, pi32, pi8]
    L.ioc_get_ties.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ioc_align_set_pool.argtypes = [vp, i32, C.c_char_p, C.POINTER(C.c_int64)]
    L.ioc_align_pairs.argtypes = [vp, i32, C.POINTER(AlnPair), i32, i32, i32, i32, pi32, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double)]
    L.ioc_dist_unique_id.argtypes = [pu8]
    L.ioc_dist_init.argtypes = [vp, pu8, i32, i32]
    L.ioc_dist_shutdown.argtypes = [vp]
    L.ioc_dist_info.argtypes = [vp, pi32, pi32]
    L.ioc_dist_allgather_device.argtypes = [vp, vp, vp, i64]
    L.ioc_dist_allgatherv_device.argtypes = [vp, vp, vp, pi64, pi64, i32]
    L.ioc_dist_allgather_i64.argtypes = [vp, i64, pi64]
    L.ioc_dist_allgatherv_host.argtypes = [vp, vp, i64, vp, pi64]
    L.ioc_dist_allreduce_max.argtypes = [vp, pd]
    L.ioc_dist_barrier.argtypes = [vp]
    L.ioc_dist_merge.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(BatchView), i32, i64, pi32, pi8, pi64,
                                 C.POINTER(ClusterStats), C.POINTER(DistMergeTimes)]
    L.ioc_set_shard.argtypes = [vp, i32, i32, EXCHANGE_FN, vp]
    L.ioc_shard_exchanges.argtypes = [vp]
    L.ioc_shard_aligned_pairs.argtypes = [vp]
    L.ioc_shard_aligned_pairs.restype = C.c_int64
    L.ioc_dist_exchange.argtypes = [vp, vp, i64, i32]
    L.ioc_dist_set_shard.argtypes = [vp, i32]
    L.ioc_align_set_verdict_threshold.argtypes = [vp, C.c_double]
    _lib = L
    return L
