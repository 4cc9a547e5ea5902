#!/usr/bin/env python3
"""Debug aid: a few alignment pairs through the GPU aligner against the host aligner.  tools/dbg_align_v2.py LEN [NPAIRS] [seed]"""
import ctypes as C, random, sys
sys.path.insert(0, ".")
from isonclust2_amd import _lib, api
L = int(sys.argv[1]); NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = random.Random(seed)
def mut(s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3: out.append(rng.choice(b"ACGT"))
        elif x < 2 * rate / 3: continue
        elif x < rate: out.append(ch); out.append(rng.choice(b"ACGT"))
        else: out.append(ch)
    return bytes(out)
seqs, pairs = [], []
for t in range(NP):
    a = bytes(rng.choice(b"ACGT") for _ in range(max(1, L + rng.randint(-L // 20, L // 20))))
    b = mut(a, 0.12) if t % 3 != 2 else bytes(rng.choice(b"ACGT") for _ in range(L))
    seqs += [a, b]; pairs.append((2 * t, 2 * t + 1, t % 2, 0.1 + 0.02 * (t % 4)))
print("context", flush=True)
ctx = api.Context(0)
ctx.align_set_pool(seqs)
print("aligning", NP, "pairs of", L, flush=True)
sc, win, ratio = ctx.align_pairs(pairs, 11)
print("device done", ctx.timings()["ms_align_fwd"], ctx.timings()["ms_align_trace"], flush=True)
Lh = _lib.load()
comp = {65: 84, 67: 71, 71: 67, 84: 65}
bad = 0
for i, (qi, ri, rc, e) in enumerate(pairs):
    q, r = seqs[qi], seqs[ri]
    if rc: r = bytes(comp[ch] for ch in reversed(r))
    cap = len(q) + len(r) + 2
    buf = C.create_string_buffer(cap); s = C.c_int32()
    n = Lh.ioc_host_align(q, len(q), r, len(r), 2, -2, Lh.ioc_host_gap_open(e), 1, buf, cap, C.byref(s))
    hr = Lh.ioc_host_aln_ratio(buf, n, e, len(q), 11)
    if s.value != sc[i] or hr != ratio[i]:
        bad += 1
        if bad < 6: print("MISMATCH pair", i, len(q), len(r), "host", s.value, hr, "device", sc[i], ratio[i], flush=True)
print("pairs", NP, "mismatches", bad, flush=True)
