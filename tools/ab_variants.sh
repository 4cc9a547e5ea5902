#!/bin/bash
# developer A/B on the GPU box: bench.py (fast mode) with the library variants named on the command line
# (isonclust2_amd/_variants/lib_<name>.so, built with make OUT=... EXTRA=-D...; "base" = the shipped library)
for v in "$@"; do
  if [ $v = base ]; then unset IOC_LIB; else export IOC_LIB=$PWD/isonclust2_amd/_variants/lib_$v.so; fi
  timeout -k 10 200 python bench.py --mode fast --steps 20 --warmup 3 --no-cli --no-core --no-merge --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<P
import json; d=json.load(open("gpurun_out/ab_$v.json")); print("$v", d["ms_per_step"], d["phase_ms"], d["roofline"]["kernel_ms"], d.get("parity") or d.get("golden_parity"))
P
done
