#!/bin/bash
# Every kernel of ONE resident step (the run's last): start offset, duration, the idle gap before it.  tools/step_timeline.sh TAG [sahlin|fast]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
MODE=${2:-sahlin}
OUT=gpurun_out/timeline_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-core > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_distinct_radix" in r["Kernel_Name"]]
# the last two starts of a step bracket the last complete step
a, b = st[-2], st[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open("$OUT/step.txt", "w")
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    name = r["Kernel_Name"].split("(")[0][-48:]
    print(f'{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  gap {gap:8.1f}  {name}', file=out)
    busy += (e - s)
    prev_end = max(prev_end, e)
print(f'step: {(int(rows[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us from its first kernel to the next step\\'s, kernels {busy / 1e3:.1f} us (overlapping ones counted twice)', file=out)
PY
cat $OUT/step.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
