#!/bin/bash
# Developer aid: ONE level-0 merge of configs[4] (two 31 250-read leaves, sahlin, consensus 20,100,400) with IOC_TRACE laps.
cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/mt; rm -rf $D; mkdir -p $D
python3 - <<PY
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
with open("/tmp/mt/r.fq", "wb") as f:
    for b in range(2):
        rs = synth.generate(31250, 1500, 2000, 10, 21, seed=1000 + b, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d_%d\n" % (b, i) + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 62500 -M 31250 -g 20 -c 100 -P 400 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
for b in 0 1; do $CLI cluster -l $D/s/batches/isONbatch_$b.cer -o $D/c$b.cer -x sahlin > /dev/null 2>&1 || exit 1; done
ls -la $D/c0.cer $D/c1.cer
( time IOC_TRACE=1 ISONCLUST2_STATS_JSON=1 $CLI cluster -l $D/c0.cer -r $D/c1.cer -o $D/m.cer -x sahlin ) 2>&1 | grep -v "consensus pass from\|deferred:\|candidate tables\|aligner v2\|^\[ioc\]   " | tail -40
( time ISONCLUST2_STATS_JSON=1 $CLI cluster -l $D/c0.cer -r $D/c1.cer -o $D/m2.cer -x sahlin ) 2>&1 | tail -4
$CLI serve stop
