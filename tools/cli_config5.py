#!/usr/bin/env python3
"""Runs the isONclust2-hip CLI end to end on BASELINE.json configs[4]'s shape ("config 5" in SURVEY.md §8(d): 2 M reads /
4 Gb in 64 batches of 31 250 reads x 2 kb, sahlin mode, consensus on), or a slice of it: sort -> cluster each batch ->
merge -> dump, through the files, on `--gpus N` GPUs of one node (default 1).  The way the reference's external pipeline runs the
steps (README.md:105-117, main.cpp:262-275: batch b -> worker b mod N): the `cluster` jobs of the leaves are independent
processes, the merges form a binary tree whose levels are independent processes again; job j of a stage runs on GPU j mod N
(ISONCLUST2_DEVICE, read by the command line), `--jobs J` processes at a time PER GPU share a card (J = 1 and `--fold` give
round 2's first form: one process at a time, merges folded left to right).  With more GPUs asked for than the node shows, the
jobs are mapped onto the visible ones (a rehearsal of the scheduling: the output says so).  Prints wall times and what the domain offers as size-independent
checks (every read assigned exactly once, clusters ~ the generator's transcripts).
    tools/cli_config5.py NB [--per 31250] [--mode sahlin] [--cons 20,100,400] [--gpus 8] [--jobs 4] [--fold] [--gen-procs 12]"""
import argparse
import json
import multiprocessing
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, ".")
from isonclust2_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("nb", type=int, nargs="?", default=2)
ap.add_argument("--per", type=int, default=31250)
ap.add_argument("--mode", default="sahlin")
ap.add_argument("--cons", default=None, help="ConsMinSize,ConsMaxSize,ConsPeriod: consensus mode")
ap.add_argument("--jobs", type=int, default=4, help="cluster / merge processes at a time PER GPU (they share the card; keep <= 5)")
ap.add_argument("--gpus", type=int, default=1, help="GPUs of the node: job j of a stage runs on GPU j mod N (ISONCLUST2_DEVICE)")
ap.add_argument("--visible", type=int, default=0, help="GPUs this node really has, where sysfs does not say (rehearsals of --gpus N on fewer cards)")
ap.add_argument("--fold", action="store_true", help="merge left to right instead of as a binary tree")
ap.add_argument("--gen-procs", type=int, default=min(12, os.cpu_count() or 1))
ap.add_argument("--progress", default=None, help="file that receives the results so far after every stage")
ap.add_argument("--keep", action="store_true", help="keep the intermediate files of consumed steps")
a = ap.parse_args()
assert 1 <= a.jobs <= 5, "the GPU box allows 6 processes on the card at once"
assert a.gpus >= 1


def visible_gpus():
    """GPUs of this node without touching HIP: the kfd topology in sysfs (None: unknown)"""
    import glob
    n, nodes = 0, glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    try:
        for f in nodes:
            for line in open(f):
                if line.startswith("simd_count"):
                    n += int(line.split()[1]) > 0
                    break
    except OSError:          # (an ordinary user may not read the topology)
        return None
    return n


VIS = a.visible if a.visible else visible_gpus()
MAPPED = VIS is not None and 0 < VIS < a.gpus       # fewer cards than asked for: the schedule is rehearsed on what is there
CLI = os.environ.get("IOC_CLI", os.path.join("isonclust2_amd", "bin", "isONclust2-hip"))
G, LEN, TR_SEED = 1500, 2000, 11
d = tempfile.mkdtemp(prefix="ioc_cli5_")
env = dict(os.environ, ISONCLUST2_STATS_JSON="1")
t_all = time.time()


def log(msg):
    print(f"[cli_config5 {time.time() - t_all:7.1f} s] {msg}", file=sys.stderr, flush=True)


# ---- the reads: chunks of one batch's size, made in parallel from one transcriptome -------------------------------------
def gen_chunk(c):
    rs = synth.generate(a.per, G, LEN, 10, 21, seed=1000 + c, tr_seed=TR_SEED)
    p = os.path.join(d, f"part{c}.fq")
    with open(p, "wb") as f:
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d\n" % (c * a.per + i) + s + b"\n+\n" + q + b"\n")
    return p


t = time.time()
with multiprocessing.Pool(a.gen_procs) as pool:
    parts = pool.map(gen_chunk, range(a.nb))
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    for p in parts:
        with open(p, "rb") as g:
            shutil.copyfileobj(g, f, 1 << 24)
        os.unlink(p)
n_reads = a.nb * a.per
out = {"workload": f"{a.nb} x {a.per} reads of 2 kb, {a.mode}; {G} transcripts, chunk seeds 1000.., transcript seed {TR_SEED}",
       "fastq_MB": os.path.getsize(fq) / 1e6, "generate_s": time.time() - t, "jobs_per_gpu": a.jobs, "gpus": a.gpus,
       "gpus_visible": VIS, "devices_mapped_onto_visible": bool(MAPPED),
       "merge_shape": "left fold" if a.fold else "binary tree"}
log(f"reads written ({out['fastq_MB']:.0f} MB)")

stats = []


def checkpoint():
    if a.progress:
        with open(a.progress, "w") as f:
            json.dump(out, f)


def run(args, tag="", slot=0):
    t0 = time.time()
    dev = slot % a.gpus
    if MAPPED:
        dev %= VIS
    r = subprocess.run([CLI] + args, capture_output=True, text=True, env=dict(env, ISONCLUST2_DEVICE=str(dev)))
    assert r.returncode == 0, (args, r.stderr[-2000:])
    dt = time.time() - t0
    log(f"{args[0]} {tag} [gpu {dev}]: {dt:.1f} s")
    if args[0] == "cluster":
        js = [ln for ln in r.stderr.splitlines() if ln.startswith("{")]
        if js:
            stats.append(json.loads(js[-1]))
    return dt


def run_many(jobs):
    """jobs: (args, tag) of independent processes; job j on GPU j mod N, a.jobs at a time per GPU (one worker pool per GPU, so
    that a slow job holds up its own card only).  Returns (per-process seconds in job order, wall seconds)."""
    t0 = time.time()
    secs = [0.0] * len(jobs)
    per_gpu = [[j for j in range(len(jobs)) if j % a.gpus == g] for g in range(a.gpus)]

    def drain(g):
        with ThreadPoolExecutor(a.jobs) as ex:
            for j, dt in zip(per_gpu[g], ex.map(lambda j: run(jobs[j][0], jobs[j][1], slot=j), per_gpu[g])):
                secs[j] = dt

    with ThreadPoolExecutor(a.gpus) as outer:
        list(outer.map(drain, range(a.gpus)))
    return secs, time.time() - t0


sort_args = ["sort", "-B", "1000000", "-M", str(a.per)]
if a.cons:
    c = a.cons.split(",")
    sort_args += ["-g", c[0], "-c", c[1], "-P", c[2]]
    out["consensus"] = c
out["sort_s"] = run(sort_args + ["-o", os.path.join(d, "sorted"), fq])
if not a.keep:
    os.unlink(fq)
bdir = os.path.join(d, "sorted", "batches")
batches = sorted((x for x in os.listdir(bdir) if x.endswith(".cer")), key=lambda x: int(x.split("_")[1].split(".")[0]))
out["n_batches"] = len(batches)
checkpoint()

# ---- leaves ------------------------------------------------------------------------------------------------------------
leaf = [os.path.join(d, f"c{i}.cer") for i in range(len(batches))]
out["cluster_s"], out["cluster_wall_s"] = run_many(
    [(["cluster", "-l", os.path.join(bdir, b), "-o", leaf[i], "-x", a.mode], f"batch {i}") for i, b in enumerate(batches)])
if not a.keep:
    shutil.rmtree(bdir)
checkpoint()

# ---- merges ------------------------------------------------------------------------------------------------------------
out["merge_s"], out["merge_wall_s"] = [], 0.0
if a.fold:
    acc = leaf[0]
    for i in range(1, len(leaf)):
        o = os.path.join(d, f"m{i}.cer")
        out["merge_s"].append(run(["cluster", "-l", acc, "-r", leaf[i], "-o", o, "-x", a.mode], f"fold {i}"))
        if not a.keep:
            os.unlink(acc)
            os.unlink(leaf[i])
        acc = o
    out["merge_wall_s"] = sum(out["merge_s"])
else:
    level, lv = leaf, 0
    out["merge_levels"] = []
    while len(level) > 1:
        nxt, jobs, used = [], [], []
        for x in range(0, len(level) - 1, 2):
            o = os.path.join(d, f"m{lv}_{x // 2}.cer")
            jobs.append((["cluster", "-l", level[x], "-r", level[x + 1], "-o", o, "-x", a.mode], f"level {lv} pair {x // 2}"))
            used += [level[x], level[x + 1]]
            nxt.append(o)
        if len(level) % 2:
            nxt.append(level[-1])     # the odd one out moves up a level as it is (the order of the batches is kept)
        secs, wall = run_many(jobs)
        out["merge_s"] += secs
        out["merge_wall_s"] += wall
        out["merge_levels"].append({"merges": len(jobs), "wall_s": wall, "max_s": max(secs)})
        if not a.keep:
            for u in used:
                os.unlink(u)
        level, lv = nxt, lv + 1
        checkpoint()
    acc = level[0]

out["dump_s"] = run(["dump", "-i", os.path.join(d, "sorted", "sorted_reads_idx.cer"), "-o", os.path.join(d, "dump"), acc])
tsv = os.path.join(d, "dump", "clusters.tsv")
if os.path.exists(tsv):
    cl_ids, rd_ids, n = set(), set(), 0
    with open(tsv) as f:
        next(f)
        for line in f:
            p = line.rstrip("\n").split("\t")
            cl_ids.add(p[0])
            rd_ids.add(p[2])
            n += 1
    out["clusters"], out["reads_assigned"] = len(cl_ids), n
    out["every_read_assigned_once"] = n == n_reads and len(rd_ids) == n_reads
    out["transcripts_in_generator"] = G
out["pipeline_wall_s"] = out["sort_s"] + out["cluster_wall_s"] + out["merge_wall_s"] + out["dump_s"]
out["process_seconds"] = out["sort_s"] + sum(out["cluster_s"]) + sum(out["merge_s"]) + out["dump_s"]
out["total_wall_s_with_generation"] = time.time() - t_all
out["reads_per_s_pipeline"] = n_reads / out["pipeline_wall_s"]
out["cluster_stats"] = stats
print(json.dumps(out))
shutil.rmtree(d, ignore_errors=True)
