#!/bin/bash
# aligner kernels of the LAST resident sahlin step of tools/corridor_seeds.py SEED 1 (a batch with a second, small alignment round: seeds 3, 5)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/trace_seed_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/corridor_seeds.py $1 1 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
al = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("k_fwd2", "k_trace2"))]
# the last step: walk back from the end to the last index build before the aligner kernels
last = al[-1]
start = max(i for i in range(last) if rows[i]["Kernel_Name"].startswith("k_distinct_radix"))
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:last + 2]:
    if any(k in r["Kernel_Name"] for k in ("k_fwd2", "k_trace2", "k_decide_scan")):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%10.1f .. %10.1f  %9.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:48]))
PY
