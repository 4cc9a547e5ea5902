#!/usr/bin/env python3
"""Where a one-shot `cluster` process spends its time: config 2's batch through the CLI, REPS processes back to back per
variant, with IOC_TRACE laps and the ISONCLUST2_STATS_JSON phases.  tools/cli_breakdown.py MODE REPS [ENV=VAL,ENV=VAL ...]
Each further argument is one variant (a comma-separated environment).  Developer aid for VERDICT r4 item 1."""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from isonclust2_amd import synth  # noqa: E402

CLI = os.path.join("isonclust2_amd", "bin", "isONclust2-hip")
mode = sys.argv[1] if len(sys.argv) > 1 else "sahlin"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = sys.argv[3:] or [""]
full = bool(os.environ.get("BREAKDOWN_FULL"))
rs = synth.generate_config("config2", seed=1)
d = tempfile.mkdtemp(prefix="ioc_cli_")
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    for i in range(rs.n):
        s, q = rs.read(i)
        f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
subprocess.check_call([CLI, "sort", "-B", "60000", "-M", "3000", "-o", os.path.join(d, "sorted"), fq])
batch = os.path.join(d, "sorted", "batches", "isONbatch_0.cer")
if os.environ.get("BREAKDOWN_PARENT_CTX"):  # as bench.py does: the parent holds a warm context (and its arenas) while the children run
    import torch
    import bench
    from isonclust2_amd import api, pipeline
    torch.cuda.set_device(0)
    ctx = api.Context(0)
    bench.prepare(ctx, api, pipeline, synth, "config2", 1, 11, 15, 0)
    for arena in os.environ["BREAKDOWN_PARENT_CTX"].split("+"):
        if arena == "fat":
            os.environ["IOC_ALIGN_ARENA"] = "fat"
        ctx.cluster_resident()
        os.environ.pop("IOC_ALIGN_ARENA", None)
    free_b, total_b = torch.cuda.mem_get_info()
    print(f"parent context open, arenas {os.environ['BREAKDOWN_PARENT_CTX']}, VRAM in use {(total_b - free_b) / 1e9:.1f} GB", flush=True)
for v in variants:
    env = dict(os.environ, ISONCLUST2_STATS_JSON="1", IOC_TRACE="1")
    for kv in filter(None, v.split(",")):
        a, b = kv.split("=", 1)
        env[a] = b
    print(f"=== variant [{v}] mode {mode}", flush=True)
    for rep in range(reps):
        t = time.perf_counter()
        t_spawn = time.monotonic() * 1e3
        r = subprocess.run([CLI, "cluster", "-l", batch, "-o", os.path.join(d, "out.cer"), "-x", mode], capture_output=True, text=True, env=env)
        wall = (time.perf_counter() - t) * 1e3
        t_back = time.monotonic() * 1e3
        js = [l for l in r.stderr.splitlines() if l.startswith("{")]
        j = json.loads(js[-1]) if js else {"rc": r.returncode, "err": r.stderr[-300:]}
        print(f"rep {rep}: wall {wall:.1f} ms  " + " ".join(f"{k}={j[k]:.1f}" for k in ("cli_ms", "core_ms", "load_ms", "flatten_ms", "ctx_ms", "bookkeeping_ms", "save_ms") if k in j)
              + (f" before_main={j['t_begin_mono_ms'] - t_spawn:.1f} after_exit={t_back - j['t_end_mono_ms']:.1f}" if "t_begin_mono_ms" in j else "") + (f" FAILED rc={j['rc']}: {j['err']}" if "rc" in j else ""), flush=True)
        if full or rep == reps - 1:
            print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[ioc]"))[:6000], flush=True)
subprocess.call([CLI, "serve", "stop"])
subprocess.call(["rm", "-rf", d])
