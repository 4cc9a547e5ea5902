#!/bin/bash
# A/B of aligner-v2 knobs on the config-3 batch: tools/ab_v2.sh "ENV1=a ENV2=b" "ENV1=c" ...   (one bench run per argument)
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --mode sahlin --steps ${STEPS:-4} --warmup ${WARM:-1} --no-cpu-baseline --no-cli --no-core > /tmp/ab.json 2>/tmp/ab.err
  python - "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    p = d["phase_ms"]
    l = d.get("alignment_other_aligner") or {}
    print(f"{sys.argv[1]:40s} headline ({d['config'].get('aligner','')[:22]}) step {d['ms_per_step']:7.2f} ms  fwd {p['align_fwd']:7.2f}  trace {p['align_trace']:6.2f} | other step {l.get('ms_per_step',0):7.2f} fwd {l.get('align_fwd_ms',0):7.2f} trace {l.get('align_trace_ms',0):6.2f} | parity {[v[:5] for v in d['golden_parity'].values()]}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-300:], flush=True)
PY
done
