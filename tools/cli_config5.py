#!/usr/bin/env python3
"""Runs the isONclust2-hip CLI end to end on a slice of BASELINE.json configs[4]'s shape ("config 5" in SURVEY.md §8(d):
2 M reads / 4 Gb in 64 batches of 31 250 reads x 2 kb, sahlin mode, consensus on): NB batches — sort -> cluster each
batch -> fold the merges left to right -> dump — through the files, one GPU; prints wall times and what the domain
offers as size-independent checks (every read assigned exactly once, clusters = the generator's transcripts).
    tools/cli_config5.py NB [reads per batch] [mode] [ConsMinSize,ConsMaxSize,ConsPeriod]"""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from isonclust2_amd import synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 31250
mode = sys.argv[3] if len(sys.argv) > 3 else "sahlin"
cons = sys.argv[4].split(",") if len(sys.argv) > 4 else None   # ConsMinSize,ConsMaxSize,ConsPeriod: consensus mode
CLI = os.path.join("isonclust2_amd", "bin", "isONclust2-hip")
rs = synth.generate(nb * per, 1500, 2000, 10, 21, seed=11)
d = tempfile.mkdtemp(prefix="ioc_cli4_")
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
env = dict(os.environ, ISONCLUST2_STATS_JSON="1")


stats = []


def run(args):
    t = time.time()
    r = subprocess.run([CLI] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, (args, r.stderr[-2000:])
    print(f"[cli_config5] {args[0]} {os.path.basename(args[-1]) if args[0] != 'sort' else ''}: {time.time() - t:.1f} s", file=sys.stderr, flush=True)
    if args[0] == "cluster":
        js = [l for l in r.stderr.splitlines() if l.startswith("{")]
        if js:
            stats.append(json.loads(js[-1]))
    return time.time() - t


out = {"workload": f"{nb} x {per} reads of 2 kb, {mode}", "fastq_MB": os.path.getsize(fq) / 1e6}
sort_args = ["sort", "-B", "1000000", "-M", str(per)]
if cons:
    sort_args += ["-g", cons[0], "-c", cons[1], "-P", cons[2]]
    out["consensus"] = cons
out["sort_s"] = run(sort_args + ["-o", os.path.join(d, "sorted"), fq])
batches = sorted((x for x in os.listdir(os.path.join(d, "sorted", "batches")) if x.endswith(".cer")),
                 key=lambda x: int(x.split("_")[1].split(".")[0]))
out["n_batches"] = len(batches)
cl = []
for i, b in enumerate(batches):
    o = os.path.join(d, f"c{i}.cer")
    cl.append(run(["cluster", "-l", os.path.join(d, "sorted", "batches", b), "-o", o, "-x", mode]))
out["cluster_s"] = cl
mg = []
acc = os.path.join(d, "c0.cer")
for i in range(1, len(batches)):
    o = os.path.join(d, f"m{i}.cer")
    mg.append(run(["cluster", "-l", acc, "-r", os.path.join(d, f"c{i}.cer"), "-o", o, "-x", mode]))
    acc = o
out["merge_s"] = mg
out["dump_s"] = run(["dump", "-i", os.path.join(d, "sorted", "sorted_reads_idx.cer"), "-o", os.path.join(d, "dump"), acc])
tsv = os.path.join(d, "dump", "clusters.tsv")
if os.path.exists(tsv):
    ids = set()
    n = 0
    for line in open(tsv).read().splitlines()[1:]:
        ids.add(line.split("\t")[0])
        n += 1
    out["clusters"], out["reads_assigned"] = len(ids), n
    out["every_read_assigned_once"] = n == rs.n and len(set(line.split("\t")[2] for line in open(tsv).read().splitlines()[1:])) == rs.n
    out["transcripts_in_generator"] = 1500
out["total_s"] = out["sort_s"] + sum(cl) + sum(mg) + out["dump_s"]
out["cluster_stats"] = stats
if os.environ.get("IOC_CLI4_TRACE"):   # phase trace of the slowest batch, aggregated
    worst = max(range(len(cl)), key=lambda i: cl[i])
    r = subprocess.run([CLI, "cluster", "-l", os.path.join(d, "sorted", "batches", batches[worst]), "-o", os.path.join(d, "x.cer"), "-x", mode],
                       capture_output=True, text=True, env=dict(env, IOC_TRACE="1"))
    agg = {}
    for line in r.stderr.splitlines():
        if line.startswith("[ioc] ") and line.rstrip().endswith("ms") is False and " ms" in line:
            pass
        if line.startswith("[ioc] "):
            parts = line[6:].rsplit(None, 2)
            try:
                ms = float(parts[-2]) if parts[-1] == "ms" else float(line.split(" ms")[0].split()[-1])
            except Exception:  # noqa: BLE001
                continue
            name = line[6:34].strip()
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += ms
    out["slowest_batch_trace"] = {k: [v[0], round(v[1], 1)] for k, v in agg.items()}
    out["slowest_batch_all_lines"] = [l for l in r.stderr.splitlines() if l.startswith("[ioc]") or l.startswith("{")][:60]
    out["slowest_batch_verdict_lines"] = [l for l in r.stderr.splitlines() if "verdicts" in l or "alignment batch" in l or "candidate tables" in l or "candidate lists" in l or "lists sorted" in l]
print(json.dumps(out))
subprocess.call(["rm", "-rf", d])
