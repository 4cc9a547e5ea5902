"""Byte-level golden of the `.cer` batch file: a tiny non-consensus batch assembled BY HAND here from
SURVEY.md App. B and the reference's serialize() member orders — Batch `src/serialize.h:38-43`, CmdArgs `src/args.h:32-35`,
ProcSeq `src/cluster_data.h:24`, Seq `src/seq.h:63`, Minimizer `src/minimizer.h:27` — under cereal's BinaryOutputArchive
conventions (native little endian; arithmetic = raw bytes, bool 1 B, enum = int32; std::string / std::vector = u64 count +
payload; unordered_map = u64 count + (key, value) pairs; unique_ptr = u8 valid flag; shared_ptr = u32 id, MSB set on first
occurrence, then the object).

What this pins: the layout `csrc/cli/cer.cpp` CLAIMS — it writes exactly these bytes and reads them back, and a second,
larger image assembled here (two clusters, shared members, null HPC sequences, an emptied MinDB list) loads with the same
content.  What it CANNOT show: that cereal produces the same bytes — cereal is absent from the reference tree (vendor/cereal
is an empty submodule), so interchange with a reference-written file stays unverified; and consensus-mode files are
explicitly this build's own (ConsGs entries are blobs of its POA engine, refused by name when foreign)."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.environ.get("IOC_CLI") or os.path.join(ROOT, "isonclust2_amd", "bin", "isONclust2-hip")

i32 = lambda v: struct.pack("<i", v)
u32 = lambda v: struct.pack("<I", v)
u64 = lambda v: struct.pack("<Q", v)
f64 = lambda v: struct.pack("<d", v)
b8 = lambda v: struct.pack("<B", 1 if v else 0)
st = lambda s: u64(len(s)) + s


def cmd_args(in_fastq=b"a", out_folder=b"b", mode=1, k=11, w=15):
    # Verbose b | Debug b | InFastq | KmerSize BatchSize BatchMaxSeq WindowSize MinShared ConsMinSize ConsMaxSize ConsPeriod
    # MinClsSize | MinQual MappedThreshold AlignedThreshold MinFraction MinProbNoHits | BatchOutFolder | Mode   (args.h:32-35)
    out = b8(0) + b8(0) + st(in_fastq)
    for v in (k, 50000, 30000, w, 5, 50, -150, 500, 3):
        out += i32(v)
    for v in (7.0, 0.65, 0.2, 0.8, 0.1):
        out += f64(v)
    return out + st(out_folder) + i32(mode)


def seq(name, s, q, score, err):            # seq.h:63
    return st(name) + st(s) + st(q) + f64(score) + f64(err)


def proc_seq(raw, hpc, mins, rev, strand, ident):   # cluster_data.h:24; unique_ptr = flag byte + object
    out = (b8(1) + seq(*raw)) if raw else b8(0)
    out += (b8(1) + seq(*hpc)) if hpc else b8(0)
    for lst in (mins, rev):
        out += u64(len(lst)) + b"".join(u32(a) + u32(b) + u32(c) for a, b, c in lst)   # minimizer.h:27: Min Pos Index
    return out + i32(strand) + st(ident)


def golden_bytes():
    out = i32(1) + u64(2) + u64(3) + u64(4) + i32(5) + i32(1)        # BatchNr BatchStart BatchEnd BatchBases TotalReads NrCls
    out += cmd_args()                                                # SortArgs
    out += st(b"L") + st(b"") + i32(-1)                              # LeftLeaf RightLeaf Depth
    out += u64(1) + u32(9) + u64(1) + u32(0)                         # MinDB: 1 key -> [0]
    out += u64(1) + u32(0x80000001) + u64(1) + u32(0x80000002)       # Cls: 1 shared_ptr<Cluster> holding 1 shared_ptr<ProcSeq>
    out += proc_seq((b"n", b"AC", b"II", 1.5, 0.25), None, [(1, 2, 3)], [], 1, b"i")
    out += u64(1) + b8(0)                                            # ConsGs: one null graph
    return out


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        pytest.skip("command line not built")
    return CLI


def test_writer_produces_exactly_the_spelled_out_bytes(cli, tmp_path):
    p = tmp_path / "g.cer"
    subprocess.check_call([cli, "golden", str(p)])
    got = p.read_bytes()
    want = golden_bytes()
    assert got == want, (len(got), len(want), next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None))


def test_reader_takes_a_hand_assembled_batch(cli, tmp_path):
    """Two clusters; the second one's first member is the SAME ProcSeq object as an earlier pointer (id without the MSB: no
    object follows), a null RawSeq placeholder, an emptied MinDB list (UpdateMinDB leaves those, minimizer.cpp:150-152)."""
    out = i32(7) + u64(100) + u64(102) + u64(24) + i32(0) + i32(2)
    out += cmd_args(b"reads.fq", b"out", mode=0, k=13, w=20)
    out += st(b"left.cer") + st(b"right.cer") + i32(2)
    out += u64(2) + u32(5) + u64(2) + u32(0) + u32(1) + u32(0xFFFFFFFF) + u64(0)
    out += u64(2)
    out += u32(0x80000001) + u64(2)                                   # cluster 0: two members
    out += u32(0x80000002) + proc_seq((b"rep_7_0", b"ACGTACGTAC", b"IIIIIIIIII", 9.5, 0.02), (b"rep_7_0", b"ACGTACGTAC", b"IIIIIIIIII", 9.5, 0.03),
                                      [(27, 0, 0), (44, 3, 1)], [(99, 1, 0)], 1, b"r0")
    out += u32(0x80000003) + proc_seq(None, None, [], [], -1, b"r1")
    out += u32(0x80000004) + u64(2)                                   # cluster 1
    out += u32(0x80000005) + proc_seq((b"x", b"ACGTACGTACGTAC", b"IIIIIIIIIIIIII", 3.0, 0.1), None, [], [], 1, b"r2")
    out += u32(0x00000003)                                            # ... and the placeholder object of cluster 0 again, by id
    out += u64(2) + b8(0) + b8(0)
    p = tmp_path / "h.cer"
    p.write_bytes(out)
    r = subprocess.run([cli, "info", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    e = r.stderr
    assert "Batch number: 7" in e and "Batch range: [100,102]" in e and "Depth: 2" in e and "Nr clusters: 2" in e
    assert "Nr bases: 24" in e and "Minimizers in database: 2" in e
    # every truncation of it is refused with a message (never a crash, never a silent partial batch)
    for cut in (len(out) - 1, len(out) // 2, 17, 3):
        p.write_bytes(out[:cut])
        r = subprocess.run([cli, "info", str(p)], capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr.strip(), cut


def test_foreign_graph_blob_is_refused_by_name(cli, tmp_path):
    g = golden_bytes()
    foreign = g[:-1] + b8(1) + u64(8) + b"SPOAGRPH"                   # a non-null graph that is not this build's blob
    p = tmp_path / "f.cer"
    p.write_bytes(foreign)
    r = subprocess.run([cli, "info", str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "not in this build's format" in r.stderr
