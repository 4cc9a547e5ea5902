#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh OUTTAG lib1 lib2 ...   (fast-mode bench line per library -> gpurun_out/ab_OUTTAG_<lib>.json)
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
for l in "$@"; do
  name=$(basename $l .so)
  IOC_LIB=$PWD/isonclust2_amd/$l python3 bench.py --mode ${AB_MODE:-fast} --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline --no-cli --no-core > gpurun_out/ab_${tag}_${name}.json 2> gpurun_out/ab_${tag}_${name}.err
  python3 - <<PY
import json
try:
    j=json.loads(open("gpurun_out/ab_${tag}_${name}.json").read().strip().splitlines()[-1])
    print("${name}", round(j["ms_per_step"],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["phase_ms"].items()}, j.get("golden_parity",{}).get("${AB_MODE:-fast}","")[:40])
except Exception as e:
    print("${name} FAILED", e)
PY
done
