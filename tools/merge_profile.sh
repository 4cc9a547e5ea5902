#!/bin/bash
# Developer aid, runs on the GPU box: ONE merge of configs[4] (two clustered 31 250-read x 2 kb leaves, sahlin, consensus
# 20,100,400) through the command line with IOC_TRACE=1, then under rocprofv3 --kernel-trace --stats.   tools/merge_profile.sh TAG [per]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; PER=${2:-31250}
D=/tmp/ioc_mergeprof; rm -rf $D; mkdir -p $D gpurun_out/merge_$TAG
python3 - "$PER" "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
per, d = int(sys.argv[1]), sys.argv[2]
with open(d + "/r.fq", "wb") as f:
    for b in range(2):
        rs = synth.generate(per, 1500, 2000, 10, 21, seed=1000 + b, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d_%d\n" % (b, i) + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 1000000 -M $PER -g 20 -c 100 -P 400 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
$CLI cluster -l $D/s/batches/isONbatch_0.cer -o $D/c0.cer -x sahlin > /dev/null 2>&1 || exit 1
$CLI cluster -l $D/s/batches/isONbatch_1.cer -o $D/c1.cer -x sahlin > /dev/null 2>&1 || exit 1
( time IOC_TRACE=1 ISONCLUST2_STATS_JSON=1 $CLI cluster -l $D/c0.cer -r $D/c1.cer -o $D/m.cer -x sahlin ) 2> gpurun_out/merge_$TAG/trace.err || exit 1
python3 - gpurun_out/merge_$TAG/trace.err <<'PY'
import re, collections, sys
tot, cnt = collections.Counter(), collections.Counter()
for line in open(sys.argv[1]):
    m = re.match(r'\[ioc\] (.*?)\s+([0-9.]+) ms$', line.rstrip())
    if m:
        tot[m.group(1).strip()] += float(m.group(2)); cnt[m.group(1).strip()] += 1
for k, v in tot.most_common(14):
    print(f"{v:10.1f} ms  x{cnt[k]:5d}  {k}")
PY
grep "consensus phases\|POA: \|^{\|real" gpurun_out/merge_$TAG/trace.err | cut -c1-260
[ -n "${NOPROF:-}" ] && exit 0
IOC_CLI_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/merge_$TAG/prof -- $CLI cluster -l $D/c0.cer -r $D/c1.cer -o $D/m2.cer -x sahlin > gpurun_out/merge_$TAG/prof.log 2>&1
F=$(find gpurun_out/merge_$TAG/prof -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/merge_$TAG/kernel_stats.csv && head -12 gpurun_out/merge_$TAG/kernel_stats.csv | cut -c1-140
find gpurun_out/merge_$TAG/prof -name "*kernel_trace.csv" -size +40M -delete
