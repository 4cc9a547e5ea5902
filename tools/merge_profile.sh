#!/bin/bash
# Developer aid, runs on the GPU box: ONE merge of the configs[4] tree (two 31 250-read leaves, sahlin, consensus 20,100,400) through the
# command line with IOC_TRACE=1: where its seconds go.   tools/merge_profile.sh TAG
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1
D=/tmp/ioc_mergeprof; rm -rf $D; mkdir -p $D gpurun_out/merge_$TAG
python3 - "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
d = sys.argv[1]
with open(d + "/r.fq", "wb") as f:
    for c in range(2):
        rs = synth.generate(31250, 1500, 2000, 10, 21, seed=1000 + c, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d\n" % (c * 31250 + i) + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 1000000 -M 31250 -g 20 -c 100 -P 400 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
for b in 0 1; do $CLI cluster -l $D/s/batches/isONbatch_$b.cer -o $D/c$b.cer -x sahlin > /dev/null 2>&1 || exit 1; done
( time IOC_TRACE=1 ISONCLUST2_STATS_JSON=1 $CLI cluster -l $D/c0.cer -r $D/c1.cer -o $D/m.cer -x sahlin ) 2> gpurun_out/merge_$TAG/trace.err || exit 1
grep "consensus phases\|POA\|^{\|real\|deferred consensus" gpurun_out/merge_$TAG/trace.err | cut -c1-330
grep "^\[ioc\]" gpurun_out/merge_$TAG/trace.err | grep " ms" | grep -v "consensus phases\|POA\|aligner v2\|candidate tables:" | sed -E 's/\([0-9]+ candidate tables\)//' | awk '{v=$(NF-1); $NF=""; $(NF-1)=""; k=$0; s[k]+=v; n[k]++} END {for (k in s) printf "%10.1f ms %6d  %s\n", s[k], n[k], k}' | sort -rn | head -12
