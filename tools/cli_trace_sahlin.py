import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from isonclust2_amd import synth
CLI = os.path.join("isonclust2_amd", "bin", "isONclust2-hip")
rs = synth.generate_config("config2", seed=1)
d = tempfile.mkdtemp(prefix="ioc_cli_")
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
subprocess.check_call([CLI, "sort", "-B", "60000", "-M", "3000", "-o", os.path.join(d, "sorted"), fq])
batch = os.path.join(d, "sorted", "batches", "isONbatch_0.cer")
for rep in range(2):
    t = time.time()
    r = subprocess.run([CLI, "cluster", "-l", batch, "-o", os.path.join(d, "out.cer"), "-x", "sahlin"], capture_output=True, text=True, env=dict(os.environ, ISONCLUST2_STATS_JSON="1", IOC_TRACE="1"))
    print("wall", time.time() - t)
    print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[ioc]") or l.startswith("{"))[:3000])
subprocess.call(["rm", "-rf", d])
