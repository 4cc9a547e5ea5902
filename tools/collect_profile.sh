#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the HBM-traffic PMC passes for bench.py.
# Counters go in their own runs (no sys/hip traces next to --pmc), FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/profile_$1
mkdir -p "$OUT"
# default bench = sahlin headline + fast-mode leg: the trace covers every kernel of both
CMD="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cli --no-core"
FAST="python3 bench.py --mode fast --steps 5 --warmup 1 --no-cpu-baseline --no-cli --no-core"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $FAST > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $FAST > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- $FAST > "$OUT/tcc.log" 2>&1
python3 tools/summarize_pmc.py "$OUT" "$1"
tail -2 "$OUT/trace.log"
