cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/sv; rm -rf $D; mkdir -p $D
python3 - <<PY
import sys
sys.path.insert(0, ".")
from isonclust2_amd import synth
rs = synth.generate(600, 40, 1500, 10, 21, seed=5)
with open("/tmp/sv/r.fq", "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
PY
CLI=isonclust2_amd/bin/isONclust2-hip
$CLI sort -B 1000000 -M 100 -o $D/s $D/r.fq > /dev/null 2>&1 || exit 1
ISONCLUST2_SERVE=0 $CLI cluster -l $D/s/batches/isONbatch_0.cer -o $D/ref.cer -x sahlin || exit 1
export ISONCLUST2_SERVE_DIR=/tmp/svd ISONCLUST2_SERVE_SLOTS=3
for round in 1 2 3; do
  for i in 1 2 3 4 5 6 7 8; do ( $CLI cluster -l $D/s/batches/isONbatch_0.cer -o $D/o$i.cer -x sahlin; echo "rc $?" > $D/rc$i ) & done
  wait
  ok=0; for i in 1 2 3 4 5 6 7 8; do cmp -s $D/ref.cer $D/o$i.cer && grep -q "rc 0" $D/rc$i && ok=$((ok+1)); done
  echo "round $round: $ok of 8 identical to the one-shot output; sockets: $(ls /tmp/svd | grep -c sock)"
done
$CLI serve stop
