cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/st; rm -rf $D; mkdir -p $D
python3 - <<PY
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
with open("/tmp/st/r.fq", "wb") as f:
    for b in range(8):
        rs = synth.generate(31250, 1500, 2000, 10, 21, seed=1000 + b, tr_seed=11)
        for i in range(rs.n):
            s, q = rs.read(i)
            f.write(b"@r%d_%d\n" % (b, i) + s + b"\n+\n" + q + b"\n")
PY
ls -la /tmp/st/r.fq
( time IOC_TRACE=1 isonclust2_amd/bin/isONclust2-hip sort -B 62500 -M 31250 -g 20 -c 100 -P 400 -o /tmp/st/s /tmp/st/r.fq ) 2>&1 | grep -a "sort:\|real"
