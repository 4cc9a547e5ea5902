#!/bin/bash
# SQ counters of the aligner's forward kernel under the environment given: tools/pmc_fwd.sh TAG [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_fwd_$tag
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode sahlin --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-core > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "k_fwd2" not in k or "prof" in k or "ends" in k: continue
    acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k,r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k[:40]] += 1
for k, d in acc.items():
    print("$tag", k, "launches", n[k], {c: f"{v / n[k]:.4g}" for c, v in sorted(d.items())})
PY
