"""Synthetic long-read generator for tests and bench.py (SURVEY.md §8(d) workload definition).

G transcripts of uniform-random ACGT (no N: the reference's RevComp throws on it,
src/util.cpp:31-33); each read is one transcript with substitution / insertion / deletion each at
rate e/3, e = 10^(-Q/10), Q uniform in [q_lo, q_hi]; half the reads are reverse-complemented; the
quality string is Q+33 with +-2 jitter per base.  The generator, its parameters and the seed are
part of the metric (CPU throughput varies 3x with read quality), so every bench line names them.
"""
from dataclasses import dataclass

import numpy as np

_ACGT = np.frombuffer(b"ACGT", np.uint8)
_COMP = np.zeros(256, np.uint8)
_COMP[list(b"ACGT")] = list(b"TGCA")


@dataclass
class ReadSet:
    seq: np.ndarray       # uint8 ASCII, all reads concatenated
    qual: np.ndarray      # uint8 ASCII (phred+33), same layout
    offs: np.ndarray      # int64 [n+1]
    transcript: np.ndarray  # int32 [n] ground-truth transcript of each read
    strand: np.ndarray    # int8 [n] +1 / -1
    tag: str

    @property
    def n(self):
        return len(self.offs) - 1

    def subset(self, idx):
        """The reads idx (in that order) as a new ReadSet."""
        idx = np.asarray(idx, np.int64)
        lens = np.diff(self.offs)[idx]
        offs = np.zeros(len(idx) + 1, np.int64)
        offs[1:] = np.cumsum(lens)
        gather = np.repeat(self.offs[:-1][idx] - offs[:-1], lens) + np.arange(offs[-1])
        return ReadSet(seq=self.seq[gather], qual=self.qual[gather], offs=offs, transcript=self.transcript[idx],
                       strand=self.strand[idx], tag=f"{self.tag}[{len(idx)} reads]")

    def read(self, i):
        a, b = int(self.offs[i]), int(self.offs[i + 1])
        return self.seq[a:b].tobytes(), self.qual[a:b].tobytes()


CONFIGS = {
    # name: (n_reads, n_transcripts, transcript_len, q_lo, q_hi)
    "config1": (500, 50, 1500, 10.0, 21.0),       # 0.75 Mb  (BASELINE.json configs[0])
    "config2": (3000, 150, 16667, 10.0, 21.0),    # 50.0 Mb  (BASELINE.json configs[1], primary)
    "config2_clean": (3000, 150, 16667, 16.0, 25.0),
    "config2_half": (1500, 75, 16667, 10.0, 21.0),   # (half of config 2's batch: an alignment round of ~400 couples)
    "config2_quarter": (750, 38, 16667, 10.0, 21.0),
    "config2_eighth": (375, 19, 16667, 10.0, 21.0),   # (an alignment round of ~100 couples: the few-couples launches)
    "tiny": (64, 8, 600, 10.0, 21.0),
    "short_dup": (1200, 120, 300, 10.0, 21.0),    # short reads, tie-prone
}


def generate(n_reads, n_transcripts, length, q_lo=10.0, q_hi=21.0, seed=1, dup_every=0,
             len_jitter=0.0, tr_seed=None):
    """dup_every=2 duplicates every second transcript (paralog-like, forces candidate ties).
    tr_seed: draw the transcripts from a generator of their own, so that chunks of one read set made with different
    `seed`s (in parallel processes) sample the same transcriptome."""
    rng = np.random.default_rng(seed)
    trng = rng if tr_seed is None else np.random.default_rng(tr_seed)
    tr = []
    for t in range(n_transcripts):
        L = length if len_jitter == 0 else max(64, int(length * (1 + len_jitter * (trng.random() * 2 - 1))))
        if dup_every and t % dup_every == 1:
            tr.append(tr[-1].copy())
        else:
            tr.append(_ACGT[trng.integers(0, 4, L)])
    seqs, quals = [], []
    which = rng.integers(0, n_transcripts, n_reads).astype(np.int32)
    strand = np.where(rng.random(n_reads) < 0.5, -1, 1).astype(np.int8)
    for i in range(n_reads):
        src = tr[which[i]]
        L = len(src)
        Q = q_lo + (q_hi - q_lo) * rng.random()
        e = 10.0 ** (-Q / 10.0)
        u = rng.random(L)
        dele = u < e / 3
        sub = (u >= e / 3) & (u < 2 * e / 3)
        ins = (u >= 2 * e / 3) & (u < e)
        base = src.copy()
        ns = int(sub.sum())
        if ns:
            # substitute with a different base
            cur = np.searchsorted(_ACGT, base[sub])
            base[sub] = _ACGT[(cur + rng.integers(1, 4, ns)) % 4]
        cnt = np.ones(L, np.int64)
        cnt[dele] = 0
        cnt[ins] = 2
        out = np.repeat(base, cnt)
        # the first copy of every inserted pair becomes a random base
        starts = np.cumsum(cnt) - cnt
        ins_pos = starts[ins]
        out[ins_pos] = _ACGT[rng.integers(0, 4, len(ins_pos))]
        if strand[i] < 0:
            out = _COMP[out[::-1]]
        q = np.clip(np.rint(Q + 33 + rng.integers(-2, 3, len(out))), 34, 126).astype(np.uint8)
        seqs.append(out)
        quals.append(q)
    offs = np.zeros(n_reads + 1, np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    tag = f"synth(n={n_reads},G={n_transcripts},L={length},Q=[{q_lo:g},{q_hi:g}],seed={seed},dup={dup_every})"
    if tr_seed is not None:
        tag = tag[:-1] + f",tr_seed={tr_seed})"
    return ReadSet(np.concatenate(seqs), np.concatenate(quals), offs, which, strand, tag)


def generate_config(name, seed=1):
    n, g, L, lo, hi = CONFIGS[name]
    dup = 2 if name == "short_dup" else 0
    rs = generate(n, g, L, lo, hi, seed=seed, dup_every=dup)
    rs.tag = f"{name}:{rs.tag}"
    return rs
