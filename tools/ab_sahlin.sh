#!/bin/bash
# developer A/B on the GPU box: bench.py (sahlin headline) with the library variants named on the command line
for v in "$@"; do
  if [ $v = base ]; then unset IOC_LIB; else export IOC_LIB=$PWD/isonclust2_amd/_variants/lib_$v.so; fi
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cli --no-core --no-merge --no-cpu-baseline > gpurun_out/abs_$v.json 2> gpurun_out/abs_$v.err
  python - <<P
import json; d=json.load(open("gpurun_out/abs_$v.json")); print("$v", round(d["ms_per_step"],2), round(d["phase_ms"]["align_fwd"],2), round(d["phase_ms"]["align_trace"],2), d.get("golden_parity"))
P
done
