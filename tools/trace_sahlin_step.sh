#!/bin/bash
# aligner kernels of ONE sahlin-mode step (the last of the run): name, start offset, duration in us
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/trace_sahlin_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode sahlin --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-core > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fw = [i for i, r in enumerate(rows) if "k_fwd2(" in r["Kernel_Name"] or r["Kernel_Name"].endswith("k_fwd2")]
a = fw[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open("$OUT/step.txt", "w")
for r in rows[max(0, a - 4):a + 12]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:10.1f} us .. {(e - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {r["Kernel_Name"][:60]}', file=out)
PY
cat $OUT/step.txt
