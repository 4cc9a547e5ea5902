"""FNV-1a digests of cluster assignments (what the golden fixtures and bench.py's parity fields compare)."""
import numpy as np


def fnv1a(cls, strand):
    """64-bit FNV-1a over (int32 cluster id little-endian, int8 strand) of every entry in order."""
    h = 0xcbf29ce484222325
    for c, s in zip(np.asarray(cls).tolist(), np.asarray(strand).tolist()):
        for b in (c & 0xFFFFFFFF).to_bytes(4, "little") + (s & 0xFF).to_bytes(1, "little"):
            h ^= b
            h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv1a_reads(cb, n_reads=None):
    """Digest of a ClusteredBatch over read ids 0..n_reads-1 (unassigned: cluster -1, strand 0) as a hex string."""
    if n_reads is None:
        n_reads = int(cb.member_read.max()) + 1 if len(cb.member_read) else 0
        n_reads = max(n_reads, int(cb.batch_end) + 1)
    cls, strand = cb.assignments(n_reads)
    return f"{fnv1a(cls, strand):016x}"
