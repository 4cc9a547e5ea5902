#!/usr/bin/env python3
"""Runs the isONclust2-hip CLI on the config-2 workload (3000 reads / 50 Mb) and reports the two regions of
SURVEY.md §8(d): core (ClusterSortedReads-equivalent, host arrays -> assignments, incl. H2D/D2H) and cli
(whole `cluster` process incl. .cer load/save)."""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from isonclust2_amd import synth  # noqa: E402

CLI = os.path.join("isonclust2_amd", "bin", "isONclust2-hip")
rs = synth.generate_config("config2", seed=1)
d = tempfile.mkdtemp(prefix="ioc_cli_")
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
t = time.time()
subprocess.check_call([CLI, "sort", "-B", "60000", "-M", "3000", "-o", os.path.join(d, "sorted"), fq])
t_sort = time.time() - t
batch = os.path.join(d, "sorted", "batches", "isONbatch_0.cer")
res = []
for rep in range(3):
    t = time.time()
    r = subprocess.run([CLI, "cluster", "-l", batch, "-o", os.path.join(d, "out.cer"), "-x", "fast"], capture_output=True,
                       text=True, env=dict(os.environ, ISONCLUST2_STATS_JSON="1"))
    wall = time.time() - t
    assert r.returncode == 0, r.stderr
    j = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    j["process_wall_ms"] = wall * 1e3
    res.append(j)
print(json.dumps({"workload": rs.tag, "sort_s": t_sort, "batch_cer_MB": os.path.getsize(batch) / 1e6,
                  "out_cer_MB": os.path.getsize(os.path.join(d, "out.cer")) / 1e6, "cluster_runs": res}))
subprocess.call(["rm", "-rf", d])
