#!/bin/bash
# Developer aid, runs on the GPU box: tools/poa_batch_bench.py under each library variant of isonclust2_amd/variants
# (built with `make OBJDIR=... OUT=isonclust2_amd/variants/NAME/libioc.so EXTRA=...`) and the default build.   tools/poa_variants.sh TAG [G] [LEN] [DEPTH] [ROUNDS]
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/poa_variants_$TAG.txt; : > $O
for lib in default isonclust2_amd/variants/*/libioc.so; do
  echo "== $lib" >> $O
  if [ "$lib" = default ]; then IOC_TRACE=1 timeout -k 10 300 python3 tools/poa_batch_bench.py "$@" 2>&1 | grep -v "^\[ioc\] consensus\|amdgpu.ids" | tail -12 >> $O
  else IOC_LIB=$lib IOC_TRACE=1 timeout -k 10 300 python3 tools/poa_batch_bench.py "$@" 2>&1 | grep -v "^\[ioc\] consensus\|amdgpu.ids" | tail -12 >> $O; fi
done
cat $O | cut -c1-250
