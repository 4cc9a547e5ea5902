#!/usr/bin/env python3
"""Harness of tools/micro/proc_cost: runs of one-shot GPU processes back to back (or with a pause), time before main, inside,
and after _exit.  tools/micro/proc_cost.py → table on stdout."""
import json
import os
import subprocess
import sys
import time

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proc_cost.bin")


def run(args, reps, pause=0.0, env=None, label=""):
    rows = []
    for _ in range(reps):
        t = time.monotonic() * 1e3
        r = subprocess.run([BIN] + [str(x) for x in args], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
        t2 = time.monotonic() * 1e3
        if r.returncode != 0:
            print(label, "FAILED", r.returncode, r.stderr[-200:])
            return
        j = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append((t2 - t, j["t_begin_mono_ms"] - t, j["init"], j["stream"], j["malloc"], j["touch"], j["host"], j["free_reset"], t2 - j["t_end_mono_ms"]))
        if pause:
            time.sleep(pause)
    print(f"--- {label}: args {args} pause {pause} env {env}")
    print("   wall  before   init  stream malloc  touch   host  free  after_exit")
    for r in rows:
        print(" ".join(f"{x:7.1f}" for x in r), flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    run([0, 0], reps, label="bare runtime")
    run([0, 0], reps, pause=0.5, label="bare runtime, 0.5 s apart")
    run([0, 0], reps, env={"ROCR_VISIBLE_DEVICES": "0"}, label="bare, ROCR_VISIBLE_DEVICES=0")
    run([0, 0, "exit"], reps, label="bare runtime, exit() instead of _exit")
    run([8192, 0], reps, label="8 GB VRAM touched")
    run([8192, 0], reps, pause=0.5, label="8 GB VRAM touched, 0.5 s apart")
    run([8192, 0, "free"], reps, label="8 GB VRAM touched, hipFree before exit")
    run([8192, 0, "free", "reset"], reps, label="8 GB VRAM, hipFree + hipDeviceReset")
    run([2048, 0], reps, label="2 GB VRAM touched")
    run([0, 1536], reps, label="1.5 GB host touched")
    run([0, 1536, "free"], reps, label="1.5 GB host touched, freed")
    run([8192, 1536], reps, label="8 GB VRAM + 1.5 GB host")
    run([0, 0], reps, label="bare runtime again")
