#!/bin/bash
# Runs on the GPU box: SQ counters of the alignment kernels, version 2 (k_fwd2 / k_trace2) against version 1.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_align2
mkdir -p "$OUT"
CMD="python3 bench.py --mode sahlin --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-core"
for v in v2 v1; do
  if [ $v = v1 ]; then export IOC_ALIGN_V1=1; else unset IOC_ALIGN_V1; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d "$OUT/$v" -- $CMD > "$OUT/$v.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$OUT/${v}b" -- $CMD > "$OUT/${v}b.log" 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for v in ("v2","v2b","v1","v1b"):
    g=glob.glob(f"gpurun_out/pmc_align2/{v}/**/*counter_collection.csv",recursive=True)
    if not g: print(v,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(g[0])):
        k=r["Kernel_Name"]
        if "k_align_fwd" in k or "k_fwd2" in k or "k_trace2" in k or "k_align_trace" in k:
            short = "k_fwd2_ends" if "k_fwd2_ends" in k else "k_fwd2" if "k_fwd2" in k else "k_trace2" if "k_trace2" in k else "k_align_trace" if "k_align_trace" in k else "k_align_fwd"
            agg[short][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,d in agg.items():
        print(v,k,{a:f"{b:.3e}" for a,b in d.items()}, flush=True)
PY
