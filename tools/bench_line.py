#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line on stdin (development aid)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f ms %.3f pcg/it %.2f pipeline %.4f (first %.3f) hot us %.1f roofline %.3f" % (
    d["value"], d["ms_per_step"], d["config"]["pcg_iterations_per_lm_iteration"], d["pipeline"]["seconds"],
    d["pipeline"]["first_run_in_process_seconds"], d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["frac"]))
