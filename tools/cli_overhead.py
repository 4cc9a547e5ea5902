#!/usr/bin/env python3
"""The `cluster` process on config 2's batch, three times per mode, one-shot and through the resident worker: wall time and the
phases ISONCLUST2_STATS_JSON reports (bench.py's `cli` region on its own).  tools/cli_overhead.py [fast sahlin]"""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from isonclust2_amd import synth  # noqa: E402

rs = synth.generate_config("config2", seed=1)
r = bench.cli_region(rs, sys.argv[1:] or ["sahlin", "fast"], runs=3)
print(json.dumps(r, indent=1))
