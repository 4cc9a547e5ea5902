#!/bin/bash
# Developer check, runs on the GPU box: `sort` writes the same files whatever the number of batch-writer threads.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/ioc_sortchk; rm -rf $D; mkdir -p $D
python3 - "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
d = sys.argv[1]
rs = synth.generate(6000, 200, 1200, 8, 21, seed=5, tr_seed=3)
with open(d + "/r.fq", "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
# (a batch file records the sort's arguments, the output folder among them: the same folder name for every run)
for t in 1 6 1; do
    rm -rf $D/out
    IOC_SORT_THREADS=$t $CLI sort -B 1000000 -M 700 -g 20 -c 100 -P 400 -o $D/out $D/r.fq > /dev/null 2>&1 || exit 1
    rm -rf $D/run_$t; mv $D/out $D/run_$t
done
diff -r $D/run_1 $D/run_6 > /dev/null && echo "sort: $(ls $D/run_1/batches | wc -l) batches identical with 1 and 6 writer threads" || { diff -rq $D/run_1 $D/run_6 | head -3; echo "SORT OUTPUT DIFFERS"; exit 1; }
