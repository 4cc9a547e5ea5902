#!/bin/bash
# kernel-by-kernel timeline of ONE fast-mode step (the last of the run): name, start offset and duration in us
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/trace_fast_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode fast --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-core > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the steps: every k_distinct_radix starts an index build = a step; take the last complete one that is followed by another... the 4th of 5 (warmup 1 + 3 steps + 1 untimed)
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_distinct_radix")]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open("$OUT/step.txt", "w")
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    line = f'{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r["Kernel_Name"][:70]}'
    print(line, file=out)
    prev_end = e
print(f"step: {(prev_end - t0) / 1e3:.1f} us, {b - a} launches", file=out)
PY
cat $OUT/step.txt
