#!/bin/bash
# Runs on the GPU box: SQ counters of the alignment kernels, 32-bit forward pass against the packed one.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_align
mkdir -p "$OUT"
CMD="python3 bench.py --mode sahlin --steps 1 --warmup 0 --no-cpu-baseline"
for v in packed plain; do
  if [ $v = packed ]; then export IOC_ALIGN_PACKED=1; else unset IOC_ALIGN_PACKED; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d "$OUT/$v" -- $CMD > "$OUT/$v.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d "$OUT/${v}2" -- $CMD > "$OUT/${v}2.log" 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for v in ("packed","packed2","plain","plain2"):
    g=glob.glob(f"gpurun_out/pmc_align/{v}/**/*counter_collection.csv",recursive=True)
    if not g: print(v,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(g[0])):
        k=r["Kernel_Name"]
        if "k_align_fwd" in k: agg[k.split("(")[1][:30] if False else k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,d in agg.items():
        print(v,k[:50],{a:f"{b:.3e}" for a,b in d.items()})
PY
