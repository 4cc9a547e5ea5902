"""The isONclust2-hip command line (isonclust2_amd/bin): CPU part — the .cer round trip of its own
writer/reader, error paths that need no GPU — and, under -m gpu, the whole sort -> cluster -> merge ->
dump pipeline on a synthetic FASTQ compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from isonclust2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.environ.get("IOC_CLI") or os.path.join(ROOT, "isonclust2_amd", "bin", "isONclust2-hip")


def run(*args, **kw):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600, **kw)


def test_cer_roundtrip_selftest(tmp_path):
    r = run("selftest", str(tmp_path / "t.cer"))
    assert r.returncode == 0 and "selftest ok" in r.stderr


def test_error_paths_exit_1(tmp_path):
    r = run("info", str(tmp_path / "missing.cer"))
    assert r.returncode == 1 and "Failed to load batch" in r.stderr
    r = run("cluster", "-o", str(tmp_path / "o.cer"))
    assert r.returncode == 1 and "left input batch is mandatory" in r.stderr
    r = run("sort", "-k", "9", str(tmp_path / "x.fq"))
    assert r.returncode == 1 and "kmer size" in r.stderr.lower()
    assert run("version").returncode == 0


def test_large_archive_loads_as_views(tmp_path):
    """A batch file above the 32 MB at which the loader pre-populates its mapping on several threads (the large fields of a
    loaded record are views of that mapping, cer.hpp): `info` must load it and the round trip through `golden`-sized fields
    must keep every byte — here a 2 x 17 MB record written by hand in the App. B layout."""
    import struct
    u64 = lambda v: struct.pack("<Q", v)
    i32 = lambda v: struct.pack("<i", v)
    u32 = lambda v: struct.pack("<I", v)
    f64 = lambda v: struct.pack("<d", v)
    st = lambda b: u64(len(b)) + b
    n = 17 << 20
    seq, qual = (b"ACGT" * (n // 4)), bytes([33 + (i % 40) for i in range(4096)]) * (n // 4096)
    args = (b"\x00\x00" + st(b"") + b"".join(i32(v) for v in (11, 50000, 30000, 15, 5, 50, -150, 500, 3))
            + b"".join(f64(v) for v in (7.0, 0.65, 0.2, 0.8, 0.1)) + st(b"out") + i32(1))
    rec = (u32(0x80000002) + b"\x01" + st(b"r0") + st(seq) + st(qual) + f64(5.0) + f64(0.05)
           + b"\x00" + u64(2) + b"".join(u32(v) for v in (7, 1, 0, 9, 5, 1)) + u64(0) + i32(1) + st(b"r0"))
    img = (i32(0) + u64(0) + u64(0) + u64(n) + i32(1) + i32(1) + args + st(b"") + st(b"") + i32(-1) + u64(0)
           + u64(1) + u32(0x80000001) + u64(1) + rec + u64(0))
    p = tmp_path / "big.cer"
    p.write_bytes(img)
    r = run("info", str(p))
    assert r.returncode == 0 and "Nr clusters: 1" in r.stderr, r.stderr
    # a truncated copy is refused, not read past its end
    (tmp_path / "cut.cer").write_bytes(img[:len(img) // 2])
    r = run("info", str(tmp_path / "cut.cer"))
    assert r.returncode == 1 and "truncated or corrupt" in r.stderr


def _short_dir():
    import tempfile
    return tempfile.mkdtemp(prefix="iocs", dir="/tmp")


def test_served_cluster_reports_like_the_one_shot_command(tmp_path):
    """`cluster` goes through a resident worker (serve): the caller's stderr and exit code must be the one-shot command's.
    Without a GPU both end in the same error; the worker survives the failed job and `serve stop` ends it."""
    srv = _short_dir()   # (a unix socket's path holds 107 characters: pytest's tmp_path can be longer than that)
    env = dict(os.environ, ISONCLUST2_SERVE_DIR=srv, ISONCLUST2_SERVE_IDLE_S="20")
    g = tmp_path / "g.cer"
    assert run("golden", str(g)).returncode == 0
    served = [run("cluster", "-l", str(g), "-o", str(tmp_path / "o.cer"), "-x", "fast", env=env) for _ in range(2)]
    direct = run("cluster", "-l", str(g), "-o", str(tmp_path / "o2.cer"), "-x", "fast", env=dict(env, ISONCLUST2_SERVE="0"))
    for r in served:
        assert r.returncode == direct.returncode
        assert direct.stderr.strip() in r.stderr          # (the one-shot process leaves at its first error exit, whichever thread hits it)
    # relative paths are resolved against the CALLER's directory
    r = subprocess.run([CLI, "cluster", "-l", "nope.cer", "-o", "o.cer", "-x", "fast"], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 1 and "Failed to load batch nope.cer" in r.stderr
    socks = [f for f in os.listdir(srv) if f.endswith(".sock")]
    assert len(socks) == 1, socks
    r = run("serve", "stop", env=env)
    assert "1 worker(s) stopped" in r.stderr
    assert not [f for f in os.listdir(srv) if f.endswith(".sock")]
    import shutil
    shutil.rmtree(srv, ignore_errors=True)


def test_dump_reads_the_fastq_out_of_a_mapping(tmp_path):
    """`dump` without a GPU: the golden batch (one cluster whose record carries the read id "i"), a hand-written sorted_reads_idx.cer
    and a FASTQ with that read, another one, a record of the other kind of line ending at the end: the cluster's file holds the
    read's four lines (src/output.cpp:225-275).  Runs under the sanitizer pass too (mapping + gather writer)."""
    import struct
    g = tmp_path / "g.cer"
    assert run("golden", str(g)).returncode == 0
    fq = tmp_path / "sorted_reads.fastq"
    fq.write_bytes(b"@other\nACGT\n+\nIIII\n@i\nGATTACA\n+\nABCDEFG\n@last\nAC\n+\nII")
    idx = tmp_path / "sorted_reads_idx.cer"
    idx.write_bytes(struct.pack("<Q", len(str(fq))) + str(fq).encode())
    r = run("dump", "-i", str(idx), "-o", str(tmp_path / "dump"), str(g))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "dump" / "cluster_fastq" / "0.fq").read_bytes() == b"@i\nGATTACA\n+\nABCDEFG\n"
    assert (tmp_path / "dump" / "clusters.tsv").read_text().splitlines() == ["ClusterId\tStrand\tRead", "0\t1\ti"]
    assert "@cluster_0 origin=n:1 length=2 size=0" in (tmp_path / "dump" / "cluster_cons.fq").read_text()


def test_idle_worker_leaves_by_itself(tmp_path):
    """A worker without a job for ISONCLUST2_SERVE_IDLE_S seconds ends and takes its socket with it (no GPU needed: the job
    fails at the context, the worker is up all the same)."""
    import shutil
    import time
    srv = _short_dir()
    env = dict(os.environ, ISONCLUST2_SERVE_DIR=srv, ISONCLUST2_SERVE_IDLE_S="0.3")
    g = tmp_path / "g.cer"
    assert run("golden", str(g)).returncode == 0
    run("cluster", "-l", str(g), "-o", str(tmp_path / "o.cer"), "-x", "fast", env=env)
    assert [f for f in os.listdir(srv) if f.endswith(".sock")]
    t0 = time.time()
    while [f for f in os.listdir(srv) if f.endswith(".sock")] and time.time() - t0 < 10:
        time.sleep(0.1)
    assert not [f for f in os.listdir(srv) if f.endswith(".sock")]
    assert "0 worker(s) stopped" in run("serve", "stop", env=env).stderr
    shutil.rmtree(srv, ignore_errors=True)


def _write_fastq(rs, path):
    with open(path, "wb") as f:
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d extra words\n" % i + s + b"\n+\n" + q + b"\n")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fast", "sahlin"])
def test_sort_cluster_merge_dump_matches_oracle(tmp_path, mode):
    from oracle import pyoracle as po
    rs = synth.generate(360, 30, 700, 9, 21, seed=21) if mode == "fast" else synth.generate(160, 16, 450, 9, 20, seed=22)
    half = rs.n // 2
    fq = tmp_path / "reads.fq"
    _write_fastq(rs, fq)
    out = tmp_path / "sorted"
    r = run("sort", "-v", "-B", "1000000", "-M", str(half), "-o", str(out), str(fq))
    assert r.returncode == 0, r.stderr
    b0, b1 = out / "batches" / "isONbatch_0.cer", out / "batches" / "isONbatch_1.cer"
    assert b0.exists() and b1.exists()
    for b, o in ((b0, "c0.cer"), (b1, "c1.cer")):
        r = run("cluster", "-l", str(b), "-o", str(tmp_path / o), "-x", mode)
        assert r.returncode == 0, r.stderr
    r = run("cluster", "-v", "-l", str(tmp_path / "c0.cer"), "-r", str(tmp_path / "c1.cer"), "-o", str(tmp_path / "m.cer"),
            "-x", mode, env=dict(os.environ, ISONCLUST2_STATS_JSON="1"))
    assert r.returncode == 0, r.stderr
    assert "core_ms" in r.stderr
    r = run("dump", "-i", str(out / "sorted_reads_idx.cer"), "-o", str(tmp_path / "dump"), str(tmp_path / "m.cer"))
    assert r.returncode == 0, r.stderr

    # ---- oracle: same global sort, same two batches, same fold ----
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    order, score, _ = R.order()
    p = po.default_params(11, 15)
    A, B = po.Batch(R, 0, half - 1, p, 0), po.Batch(R, half, rs.n - 1, p, 1)
    A.cluster(mode=mode)      # (sahlin: the oracle aligns with its own scalar aligner)
    B.cluster(mode=mode)
    A.cluster(right=B, mode=mode)
    ocl, ost = A.assignments(rs.n)
    # the dump sorts clusters by size (unstable std::sort, cluster.cpp:570-580): compare partitions + strands
    got = {}
    for line in open(tmp_path / "dump" / "clusters.tsv").read().splitlines()[1:]:
        c, st, name = line.split("\t")
        got[int(name[1:])] = (int(c), int(st))
    assigned = np.nonzero(ocl >= 0)[0]
    assert sorted(got) == assigned.tolist()
    m = {}
    for i in assigned:
        assert got[int(i)][1] == ost[i]
        assert m.setdefault(int(ocl[i]), got[int(i)][0]) == got[int(i)][0]
    assert len(set(m.values())) == len(m)
    # sorted_reads.fastq holds the score >= 0 reads in score order; scores.tsv every read
    names = [l[1:] for l in open(out / "sorted_reads.fastq").read().splitlines()[0::4]]
    assert names == [f"r{i}" for i, s in zip(order, score) if s >= 0]
    info = open(tmp_path / "dump" / "clusters_info.tsv").read().splitlines()
    assert len(info) - 1 == A.n_clusters()
    # cluster_fastq/<id>.fq: every read of the cluster once, in the order of the sorted FASTQ, on the cluster's strand
    # (src/output.cpp:225-275: sequence reverse-complemented and qualities reversed for MatchStrand -1)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    pos = {nm: k for k, nm in enumerate(names)}
    seen = 0
    for c_id in set(c for c, _ in got.values()):
        lines = open(tmp_path / "dump" / "cluster_fastq" / f"{c_id}.fq", "rb").read().split(b"\n")
        assert lines[-1] == b"" and (len(lines) - 1) % 4 == 0
        recs = [lines[k:k + 4] for k in range(0, len(lines) - 1, 4)]
        ids = [int(r[0][2:]) for r in recs]
        assert sorted(ids) == sorted(i for i, (c, _) in got.items() if c == c_id)
        assert [pos[f"r{i}"] for i in ids] == sorted(pos[f"r{i}"] for i in ids)
        for r, i in zip(recs, ids):
            sq, ql = rs.read(i)
            if got[i][1] == -1:
                sq, ql = sq.translate(comp)[::-1], ql[::-1]
            assert r[1] == sq and r[3] == ql and r[2] == b"+"
            seen += 1
    assert seen == len(got)


@pytest.mark.gpu
def test_consensus_mode_sort_cluster_merge_dump(tmp_path):
    """`sort -g 3 -c 8 -P 400` freezes the consensus parameters; `cluster` then keeps one POA graph per cluster
    (this build's engine, graphs carried in the .cer files), replaces representatives by consensus sequences
    and updates the MinDB; the merge runs with the Depth != -1 rules.  The oracle runs the same three steps with ITS OWN
    scalar POA (oracle/poa_oracle.cpp) behind its consensus hook — nothing of the product on that side —, the graphs of
    step 2 serving as the right batch's in the merge.  Parity with the oracle's POA; spoa itself is unpinned."""
    import ctypes as C
    from isonclust2_amd import api
    from oracle import pyoracle as po
    rs = synth.generate(200, 6, 800, 12, 21, seed=31)
    half = rs.n // 2
    fq = tmp_path / "reads.fq"
    _write_fastq(rs, fq)
    out = tmp_path / "sorted"
    r = run("sort", "-B", "1000000", "-M", str(half), "-g", "3", "-c", "8", "-P", "400", "-o", str(out), str(fq))
    assert r.returncode == 0, r.stderr
    b0, b1 = out / "batches" / "isONbatch_0.cer", out / "batches" / "isONbatch_1.cer"
    for b, o in ((b0, "c0.cer"), (b1, "c1.cer")):
        r = run("cluster", "-v", "-l", str(b), "-o", str(tmp_path / o), "-x", "fast")
        assert r.returncode == 0, r.stderr
        assert "Consensus invocation count" in r.stderr
    r = run("cluster", "-l", str(tmp_path / "c0.cer"), "-r", str(tmp_path / "c1.cer"), "-o", str(tmp_path / "m.cer"), "-x", "fast")
    assert r.returncode == 0, r.stderr
    r = run("dump", "-i", str(out / "sorted_reads_idx.cer"), "-o", str(tmp_path / "dump"), str(tmp_path / "m.cer"))
    assert r.returncode == 0, r.stderr

    # ---- the oracle, its own POA behind its hook ----
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    p = po.default_params(11, 15)
    p.cons_max_size = 8
    A, B = po.Batch(R, 0, half - 1, p, 0), po.Batch(R, half, rs.n - 1, p, 1)
    ga, gb = po.OraclePoa(), po.OraclePoa()
    events = 0
    for Bo, g in ((A, ga), (B, gb)):
        po.lib().orc_set_consensus(g.ops_pointer(), 3, 400)
        try:
            events += Bo.cluster(mode="fast")["cons_invoked"]
        finally:
            po.lib().orc_set_consensus(None, 50, 500)
    assert events > 5
    # merge: left graphs = ga's side 0, right graphs = gb's side 0 presented as side 1
    gm = po.OraclePoa()
    for src, dst_side in ((ga, 0), (gb, 1)):
        for c_id in range((A if src is ga else B).n_clusters()):
            src.copy_graph_to(c_id, gm, dst_side, c_id)
    po.lib().orc_set_consensus(gm.ops_pointer(), 3, 400)
    try:
        A.cluster(right=B, mode="fast")
    finally:
        po.lib().orc_set_consensus(None, 50, 500)
    ocl, ost = A.assignments(rs.n)
    got = {}
    for line in open(tmp_path / "dump" / "clusters.tsv").read().splitlines()[1:]:
        c, st, name = line.split("\t")
        got[int(name[1:])] = (int(c), int(st))
    assigned = np.nonzero(ocl >= 0)[0]
    assert sorted(got) == assigned.tolist()
    m = {}
    for i in assigned:
        assert got[int(i)][1] == ost[i]
        assert m.setdefault(int(ocl[i]), got[int(i)][0]) == got[int(i)][0]
    assert len(set(m.values())) == len(m)
    # representatives replaced by a consensus carry the reference's name pattern in the consensus FASTQ
    cons = open(tmp_path / "dump" / "cluster_cons.fq").read()
    assert "cons_" in cons
    for g in (ga, gb, gm):
        g.close()


@pytest.mark.gpu
def test_sort_writes_the_same_batches_whatever_the_number_of_writer_threads(tmp_path):
    """`sort` extracts batch by batch on the one context and lets worker threads assemble and write the batch files
    (IOC_SORT_THREADS): one thread and four must leave the same files.  (A batch file records the output folder: the
    same folder name for both runs.)"""
    import filecmp
    import shutil
    rs = synth.generate(900, 40, 700, 9, 21, seed=77)
    fq = tmp_path / "reads.fq"
    _write_fastq(rs, fq)
    out = tmp_path / "sorted"
    kept = {}
    for nt in ("1", "4"):
        r = run("sort", "-B", "1000000", "-M", "128", "-o", str(out), str(fq), env=dict(os.environ, IOC_SORT_THREADS=nt))
        assert r.returncode == 0, r.stderr
        kept[nt] = tmp_path / ("run" + nt)
        shutil.move(str(out), str(kept[nt]))
    names = sorted(os.listdir(kept["1"] / "batches"))
    assert len(names) == 8 and names == sorted(os.listdir(kept["4"] / "batches"))
    for sub in [""] + ["batches"]:
        cmp = filecmp.dircmp(kept["1"] / sub, kept["4"] / sub)
        match, mismatch, errors = filecmp.cmpfiles(kept["1"] / sub, kept["4"] / sub, cmp.common_files, shallow=False)
        assert not mismatch and not errors and len(match) == len(cmp.common_files) and not cmp.left_only and not cmp.right_only


@pytest.mark.gpu
def test_resident_worker_writes_the_one_shot_commands_files(tmp_path):
    """`cluster` through the resident worker (the default) and in the calling process (ISONCLUST2_SERVE=0) must leave the same
    bytes: single batches in all three modes, a merge, a consensus-mode batch; the caller's environment (ISONCLUST2_STATS_JSON)
    and its -v messages arrive; three callers at once each get a worker of their own; a failed job (a missing file) leaves the
    worker usable; `serve stop` ends them."""
    import concurrent.futures as cf
    import filecmp
    rs = synth.generate(240, 20, 600, 9, 21, seed=41)
    fq = tmp_path / "reads.fq"
    _write_fastq(rs, fq)
    out = tmp_path / "sorted"
    assert run("sort", "-B", "1000000", "-M", "120", "-o", str(out), str(fq)).returncode == 0
    assert run("sort", "-B", "1000000", "-M", "120", "-g", "3", "-c", "8", "-P", "400", "-o", str(tmp_path / "sorted_c"), str(fq)).returncode == 0
    b0, b1 = out / "batches" / "isONbatch_0.cer", out / "batches" / "isONbatch_1.cer"
    srv = _short_dir()
    env = dict(os.environ, ISONCLUST2_SERVE_DIR=srv, ISONCLUST2_SERVE_IDLE_S="60", ISONCLUST2_STATS_JSON="1")
    off = dict(env, ISONCLUST2_SERVE="0")

    def both(name, *args):
        a, b = tmp_path / (name + "_served.cer"), tmp_path / (name + "_direct.cer")
        r1 = run("cluster", *args, "-o", str(a), env=env)
        r2 = run("cluster", *args, "-o", str(b), env=off)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
        assert "core_ms" in r1.stderr and "core_ms" in r2.stderr
        strip = lambda t: [l for l in t.splitlines() if not l.startswith("{") and "Output batch written" not in l]
        assert strip(r1.stderr) == strip(r2.stderr)
        assert filecmp.cmp(a, b, shallow=False), name
        return a

    for mode in ("fast", "sahlin", "furious"):
        c0 = both("c0_" + mode, "-v", "-l", str(b0), "-x", mode)
        c1 = both("c1_" + mode, "-l", str(b1), "-x", mode)
        both("m_" + mode, "-l", str(c0), "-r", str(c1), "-x", mode)
    # consensus mode: two different batches through the same worker one after the other, then their merge (graphs in the files)
    k0 = both("cons0", "-v", "-l", str(tmp_path / "sorted_c" / "batches" / "isONbatch_0.cer"), "-x", "fast")
    k1 = both("cons1", "-l", str(tmp_path / "sorted_c" / "batches" / "isONbatch_1.cer"), "-x", "sahlin")
    both("consm", "-l", str(k0), "-r", str(k1), "-x", "sahlin")
    # a job runs under ITS caller's IOC_* / ISONCLUST2_* environment, not under an earlier caller's or the worker's own
    quiet = {k: v for k, v in env.items() if k != "ISONCLUST2_STATS_JSON"}
    r = run("cluster", "-l", str(b0), "-o", str(tmp_path / "quiet.cer"), "-x", "fast", env=quiet)
    assert r.returncode == 0 and "core_ms" not in r.stderr, r.stderr
    # a job that fails leaves its worker in place
    r = run("cluster", "-l", str(tmp_path / "missing.cer"), "-o", str(tmp_path / "x.cer"), "-x", "fast", env=env)
    assert r.returncode == 1 and "Failed to load batch" in r.stderr
    both("after_failure", "-l", str(b0), "-x", "fast")
    # three callers at once
    def job(i):
        o = tmp_path / f"par{i}.cer"
        r = run("cluster", "-l", str(b0 if i % 2 == 0 else b1), "-o", str(o), "-x", "sahlin", env=env)
        return r.returncode, o
    with cf.ThreadPoolExecutor(3) as ex:
        res = list(ex.map(job, range(3)))
    assert all(rc == 0 for rc, _ in res)
    assert filecmp.cmp(res[0][1], tmp_path / "c0_sahlin_direct.cer", shallow=False)
    assert filecmp.cmp(res[1][1], tmp_path / "c1_sahlin_direct.cer", shallow=False)
    assert filecmp.cmp(res[2][1], tmp_path / "c0_sahlin_direct.cer", shallow=False)
    r = run("serve", "stop", env=env)
    assert "worker(s) stopped" in r.stderr and not r.stderr.startswith("isONclust2-hip: 0 ")
    assert not [f for f in os.listdir(srv) if f.endswith(".sock")]
    import shutil
    shutil.rmtree(srv, ignore_errors=True)


@pytest.mark.gpu
def test_a_200_kb_read_goes_through_sort_and_cluster(tmp_path):
    """The command line on a FASTQ with one 200 kb read among ordinary ones (round 4 stopped such a batch with a capacity error):
    `sort` extracts its 50 000 minimizers on the GPU, `cluster` takes it through the long-query path of the index build; the
    partition equals the oracle's."""
    from oracle import pyoracle as po
    a, b = synth.generate(150, 15, 1200, 10, 21, seed=81), synth.generate(2, 1, 200000, 12, 18, seed=82)
    rs = synth.ReadSet(seq=np.concatenate([a.seq, b.seq]), qual=np.concatenate([a.qual, b.qual]), offs=np.concatenate([a.offs, a.offs[-1] + b.offs[1:]]),
                       transcript=np.concatenate([a.transcript, 100 + b.transcript]), strand=np.concatenate([a.strand, b.strand]), tag="long")
    fq = tmp_path / "reads.fq"
    _write_fastq(rs, fq)
    out = tmp_path / "sorted"
    assert run("sort", "-B", "10000000", "-M", str(rs.n), "-o", str(out), str(fq)).returncode == 0
    r = run("cluster", "-l", str(out / "batches" / "isONbatch_0.cer"), "-o", str(tmp_path / "c.cer"), "-x", "fast")
    assert r.returncode == 0, r.stderr
    assert run("dump", "-i", str(out / "sorted_reads_idx.cer"), "-o", str(tmp_path / "dump"), str(tmp_path / "c.cer")).returncode == 0
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(11, 15)
    B = po.Batch(R, 0, rs.n - 1, po.default_params(11, 15), 0)
    B.cluster(mode="fast")
    ocl, ost = B.assignments(rs.n)
    got = {}
    for line in open(tmp_path / "dump" / "clusters.tsv").read().splitlines()[1:]:
        c, st, name = line.split("\t")
        got[int(name[1:])] = (int(c), int(st))
    assigned = np.nonzero(ocl >= 0)[0]
    assert sorted(got) == assigned.tolist()
    m = {}
    for i in assigned:
        assert got[int(i)][1] == ost[i]
        assert m.setdefault(int(ocl[i]), got[int(i)][0]) == got[int(i)][0]
    assert len(set(m.values())) == len(m)
    assert got[rs.n - 1][0] == got[rs.n - 2][0]      # (the two reads of the 200 kb transcript share a cluster)


@pytest.mark.gpu
def test_sort_reads_fastq_like_getline(tmp_path):
    """`sort` parses its FASTQ out of a mapping: the rules are those of the line reader it replaced (src/main.cpp:109-112 reads with
    bioparser; this build always took '@name rest' / sequence / '+...' / qualities, blank lines between records skipped) — a last
    record without its newline, blank lines, descriptions behind the name all give the files of the plain input; a record whose
    qualities and bases differ in length, or that ends early, stops the command with a message."""
    import filecmp
    rs = synth.generate(60, 6, 500, 10, 21, seed=91)
    # (a batch file records the input's and the output folder's names as given: the same relative names in two directories)
    for d in ("a", "b"):
        os.mkdir(tmp_path / d)
    plain, odd = tmp_path / "a" / "reads.fq", tmp_path / "b" / "reads.fq"
    _write_fastq(rs, plain)
    with open(odd, "wb") as f:
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write((b"\n\n" if i % 7 == 3 else b"") + b"@r%d\tanother description\n" % i + s + b"\n+r%d\n" % i + q + (b"\n" if i + 1 < rs.n else b""))
    for d in ("a", "b"):
        r = subprocess.run([CLI, "sort", "-B", "1000000", "-M", "1000", "-o", "sorted", "reads.fq"], capture_output=True, text=True, cwd=str(tmp_path / d))
        assert r.returncode == 0, r.stderr
    for name in ("sorted_reads.fastq", "sorted_reads_idx.tsv", "scores.tsv", "batches/isONbatch_0.cer"):
        assert filecmp.cmp(tmp_path / "a" / "sorted" / name, tmp_path / "b" / "sorted" / name, shallow=False), name
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@x\nACGTACGTACGTACGTACGTACGTAC\n+\nIIII\n")
    r = run("sort", "-o", str(tmp_path / "s2"), str(bad))
    assert r.returncode == 1 and "Malformed fastq record: @x" in r.stderr
    bad.write_bytes(b"@x\nACGT\n+\n")
    r = run("sort", "-o", str(tmp_path / "s3"), str(bad))
    assert r.returncode == 1 and "Truncated fastq record: @x" in r.stderr
