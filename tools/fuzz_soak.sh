#!/bin/bash
# Every randomised differential soak of tools/fuzz_*.py one after the other on the GPU box, a line per run into gpurun_out/fuzz_soak_$1.txt
# (run after the round's last code commit; a failure prints the case to replay).   tools/fuzz_soak.sh TAG [scale]
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; S=${2:-1}
OUT=gpurun_out/fuzz_soak_$TAG.txt
: > $OUT
run() {
    echo "\$ python $*" >> $OUT
    timeout -k 10 900 python "$@" > gpurun_out/fuzz_last.log 2>&1
    rc=$?
    tail -2 gpurun_out/fuzz_last.log | sed 's/^/    /' >> $OUT
    [ $rc -eq 0 ] || { echo "    EXIT CODE $rc" >> $OUT; grep -i "mismatch\|bad\|error" gpurun_out/fuzz_last.log | head -5 >> $OUT; }
    echo "[soak] $* -> rc $rc"
}
run tools/fuzz_corridor.py $((40 * S)) 5001
run tools/fuzz_corridor.py $((20 * S)) 5002
run tools/fuzz_align.py $((30 * S)) 1400 5003
run tools/fuzz_verdict.py $((12 * S)) 7000 5004
run tools/fuzz_parity.py $((200 * S)) 5005
run tools/fuzz_parity.py $((60 * S)) 5006 sahlin
run tools/fuzz_parity.py $((30 * S)) 5007 furious
run tools/fuzz_poa.py $((60 * S)) 5008
run tools/fuzz_consensus.py $((60 * S)) 5009 fast
run tools/fuzz_consensus.py $((30 * S)) 5010 sahlin
run tools/fuzz_consensus_poa.py $((20 * S))
cat $OUT
