#!/usr/bin/env python3
"""Developer aid: ONE config-5 leaf (31 250 reads x 2 kb, sahlin, consensus on, the shipped POA engine) through the command line with
IOC_TRACE=1: where the seconds go.  tools/cons_leaf_trace.py [per] [cons]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from isonclust2_amd import synth
per = int(sys.argv[1]) if len(sys.argv) > 1 else 31250
cons = (sys.argv[2] if len(sys.argv) > 2 else "20,100,400").split(",")
CLI = os.path.join("isonclust2_amd", "bin", "isONclust2-hip")
d = tempfile.mkdtemp(prefix="ioc_leaf_")
rs = synth.generate(per, 1500, 2000, 10, 21, seed=1000, tr_seed=11)
fq = os.path.join(d, "r.fq")
with open(fq, "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
subprocess.check_call([CLI, "sort", "-B", str(2 * per), "-M", str(per), "-g", cons[0], "-c", cons[1], "-P", cons[2], "-o", os.path.join(d, "s"), fq],
                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t = time.time()
r = subprocess.run([CLI, "cluster", "-l", os.path.join(d, "s", "batches", "isONbatch_0.cer"), "-o", os.path.join(d, "o.cer"), "-x", "sahlin"],
                   capture_output=True, text=True, env=dict(os.environ, IOC_TRACE="1", ISONCLUST2_STATS_JSON="1"))
print("cluster wall %.2f s rc %d" % (time.time() - t, r.returncode))
lines = r.stderr.splitlines()
agg = {}
for ln in lines:
    if ln.startswith("[ioc]") and " ms" in ln and "consensus pass" not in ln:
        key = ln[5:].rsplit(None, 2)[0].strip()
        try:
            agg[key] = agg.get(key, 0.0) + float(ln.rsplit(None, 2)[1])
        except Exception:
            pass
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v:10.1f} ms  {k}")
for ln in lines:
    if "consensus phases" in ln or "POA" in ln or ln.startswith("{"):
        print(ln[:400])
subprocess.call(["rm", "-rf", d])
