#!/bin/bash
# Per-kernel register / LDS / spill figures of one object file of the build (developer aid):
#   tools/kernel_resources.sh isonclust2_amd/csrc/build/ioc_align_gpu.hip.o [name filter]
set -e
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$L/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re
cur={}
rows=[]
for line in sys.stdin:
    m=re.match(r'\s+-?\s*\.(\w+):\s+(.*)',line)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k=='name' and not v.startswith('\'') and 'args' not in cur: pass
    if k in('group_segment_fixed_size','vgpr_count','sgpr_count','vgpr_spill_count','sgpr_spill_count','private_segment_fixed_size','agpr_count'): cur[k]=v
    if k=='symbol': cur['symbol']=v
    if k=='wavefront_size': rows.append(cur); cur={}
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    if flt in r.get('symbol',''):
        print(r.get('symbol','?')[:70].ljust(70),'lds',r.get('group_segment_fixed_size'),'vgpr',r.get('vgpr_count'),'agpr',r.get('agpr_count'),'sgpr',r.get('sgpr_count'),'vspill',r.get('vgpr_spill_count'),'sspill',r.get('sgpr_spill_count'),'scratch',r.get('private_segment_fixed_size'))
" "${2:-}"
rm -rf $T
