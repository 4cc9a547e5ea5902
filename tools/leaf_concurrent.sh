#!/bin/bash
# Developer aid, runs on the GPU box: J config-5 leaves AT THE SAME TIME on one card (J processes, as tools/cli_config5.py --jobs J
# runs them), each with IOC_TRACE=1: what the device phases cost a process when it shares the card.
#   tools/leaf_concurrent.sh TAG [J] [per] [nb] [first batch]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; J=${2:-5}; PER=${3:-31250}; NB=${4:-10}; B0=${5:-5}
D=/tmp/ioc_leafconc; rm -rf $D; mkdir -p $D gpurun_out/conc_$TAG
python3 - "$PER" "$D" "$NB" <<'PY'
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
per, d, nb = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
with open(d + "/r.fq", "wb") as f:
    for b in range(nb):
        rs = synth.generate(per, 1500, 2000, 10, 21, seed=1000 + b, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d_%d\n" % (b, i) + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 1000000 -M $PER -g 20 -c 100 -P 400 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
T0=$(date +%s.%N)
for ((x = 0; x < J; ++x)); do
  b=$((B0 + x))
  ( S=$(date +%s.%N); env IOC_TRACE=1 ISONCLUST2_STATS_JSON=1 ${LEAF_ENV:-} $CLI cluster -l $D/s/batches/isONbatch_$b.cer -o $D/o$b.cer -x sahlin; E=$(date +%s.%N); echo "wall $(python3 -c "print(round($E - $S, 2))") s" >&2 ) 2> gpurun_out/conc_$TAG/leaf$b.err &
done
wait
T1=$(date +%s.%N)
echo "all $J leaves: $(python3 -c "print(round($T1 - $T0, 2))") s"
for ((x = 0; x < J; ++x)); do
  b=$((B0 + x)); echo "-- batch $b"
  grep "consensus phases\|POA: device\|POA: [0-9]\|aligner v2\|^wall" gpurun_out/conc_$TAG/leaf$b.err | tail -5 | cut -c1-230
done
