#!/bin/bash
# Developer aid, runs on the GPU box: ONE config-5 leaf (31 250 reads x 2 kb, sahlin, consensus 20,100,400) through the command
# line under rocprofv3 --kernel-trace --stats (the binary itself after `--`), then once more plain with IOC_TRACE=1.
#   tools/leaf_profile.sh TAG [per] [nb] [batch]     nb batches of `per` reads are generated and sorted, batch number `batch` is the leaf
# (the sort orders the reads by quality: a late batch holds the noisy reads — larger graphs, more alignments)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; PER=${2:-31250}; NB=${3:-1}; BI=${4:-0}
D=/tmp/ioc_leafprof; rm -rf $D; mkdir -p $D gpurun_out/leaf_$TAG
python3 - "$PER" "$D" "$NB" <<'PY'
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
per, d, nb = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
with open(d + "/r.fq", "wb") as f:
    for b in range(nb):          # (chunk seeds 1000.., transcript seed 11: tools/cli_config5.py's reads)
        rs = synth.generate(per, 1500, 2000, 10, 21, seed=1000 + b, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d_%d\n" % (b, i) + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 1000000 -M $PER -g 20 -c 100 -P 400 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
( time IOC_TRACE=1 ISONCLUST2_STATS_JSON=1 $CLI cluster -l $D/s/batches/isONbatch_$BI.cer -o $D/o.cer -x sahlin ) 2> gpurun_out/leaf_$TAG/trace.err || exit 1
grep -v "consensus pass from\|deferred:\|candidate tables\|aligner v2" gpurun_out/leaf_$TAG/trace.err | grep "consensus phases\|POA\|^{\|real" 
[ -n "${NOPROF:-}" ] && exit 0
IOC_CLI_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/leaf_$TAG/prof -- $CLI cluster -l $D/s/batches/isONbatch_$BI.cer -o $D/o2.cer -x sahlin > gpurun_out/leaf_$TAG/prof.log 2>&1
cmp $D/o.cer $D/o2.cer && echo "outputs identical"
F=$(find gpurun_out/leaf_$TAG/prof -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/leaf_$TAG/kernel_stats.csv && head -25 gpurun_out/leaf_$TAG/kernel_stats.csv | cut -c1-160
find gpurun_out/leaf_$TAG/prof -name "*kernel_trace.csv" -size +40M -delete
