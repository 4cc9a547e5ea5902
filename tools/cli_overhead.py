#!/usr/bin/env python3
"""The `cluster` process on config 2's batch, three times per mode: wall time and the phases ISONCLUST2_STATS_JSON reports
(developer aid for the CLI's fixed costs: context creation, .cer load / save).  tools/cli_overhead.py [fast sahlin]"""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from isonclust2_amd import synth  # noqa: E402

rs = synth.generate_config("config2", seed=1)
for mode in (sys.argv[1:] or ["fast", "sahlin"]):
    r = bench.cli_region(rs, mode, runs=3)
    print(mode, json.dumps({k: r[k] for k in ("process_wall_ms_all", "phases_all_runs")}))
