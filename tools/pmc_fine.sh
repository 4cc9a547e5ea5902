#!/bin/bash
# Runs on the GPU box: SQ counters of k_fwd2 / k_trace2 with and without the fine checkpoints of the diagonal blocks.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_fine
mkdir -p "$OUT"
CMD="python3 bench.py --mode sahlin --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-core"
for v in 1 0; do
  export IOC_ALIGN_V2_FINE=$v
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/f$v" -- $CMD > "$OUT/f$v.log" 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for v in ("f1","f0"):
    g=glob.glob(f"gpurun_out/pmc_fine/{v}/**/*counter_collection.csv",recursive=True)
    if not g: print(v,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(g[0])):
        k=r["Kernel_Name"]
        if "k_fwd2" in k or "k_trace2" in k:
            short = "k_fwd2_ends" if "k_fwd2_ends" in k else "k_fwd2" if "k_fwd2" in k else "k_trace2"
            agg[short][r["Counter_Name"]]+=float(r["Counter_Value"]); n[short]+=1
    for k,d in agg.items():
        print(v,k,{a:f"{b:.3e}" for a,b in d.items()}, flush=True)
PY
