#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/collect_profile.sh into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


summary = {"tag": tag, "command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (kernel trace, SQ counters); --mode fast for the TCC / HBM-traffic passes", "kernels": {}}
st = first(f"{out}/trace/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
            summary["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                   "pct": float(r["Percentage"])}
for sub in ("fetch", "write", "sq", "tcc"):
    cc = first(f"{out}/{sub}/**/*counter_collection.csv")
    if not cc:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        d = summary["kernels"].setdefault(k, {})
        for cn, val in v.items():
            d[cn + "_per_launch"] = val / max(1, len(calls[k]))
# the scoring launch = k_score_part (+ k_score_compact) in the XCD-partitioned path, k_score_t otherwise
names = [k for k in summary["kernels"] if k.startswith("k_score") or k.startswith("k_partition_mins")]
fetch_kb = sum(summary["kernels"][k].get("FETCH_SIZE_per_launch", 0.0) for k in names)
write_kb = sum(summary["kernels"][k].get("WRITE_SIZE_per_launch", 0.0) for k in names)
if names and (fetch_kb or write_kb):
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reads exactly half of the bytes
    # of a streaming read (64 B tallied per 128 B request) -> doubled; WRITE_SIZE is exact.
    traffic = {"kernel": "+".join(sorted(names)), "fetch_size_kib": fetch_kb, "write_size_kib": write_kb,
               "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
               "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; gather access widths are "
                       "uncalibrated per the guide, Infinity-Cache hits are counted in FETCH_SIZE"}
    # stamped with the build it was measured on: the sha of the library's sources (bench.py prints it beside `traffic` and says
    # when the sources have changed since)
    sys.path.insert(0, ".")
    from isonclust2_amd.digest import source_stamp
    traffic["stamp"] = {"sources_sha12": source_stamp(), "tag": tag, "collected_with": "tools/collect_profile.sh"}
    json.dump(traffic, open("profiles/k_score_traffic.json", "w"), indent=1)
    summary["k_score_traffic"] = traffic
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1, sort_keys=True)
os.makedirs("gpurun_out/profiles_export", exist_ok=True)
for f in glob.glob(f"profiles/{tag}_*") + ["profiles/k_score_traffic.json"]:
    if os.path.exists(f):
        os.system(f"cp {f} gpurun_out/profiles_export/")
print(json.dumps({k: summary["kernels"][k] for k in list(summary["kernels"])[:6]}, indent=1)[:1500])
