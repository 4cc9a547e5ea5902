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


def source_stamp():
    """sha256 (first 12 hex digits) over the library's sources (csrc/*.{hip,cpp,h,inc}, cli/, the public header): identifies the build a
    profile was taken on where git is not available (the GPU box gets a snapshot without .git)."""
    import glob
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "isonclust2_amd", "csrc", "*.*")) + glob.glob(os.path.join(root, "isonclust2_amd", "csrc", "cli", "*.*")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        if f.endswith((".hip", ".cpp", ".h", ".hpp", ".inc")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:12]
