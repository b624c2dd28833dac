#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: tools/show_bench.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print("%s: %.0f %s, %.1f ms/step, roofline %s %.1f %s = %.3f of peak, avg launch %.3f ms, timed region %.1f s"
      % (d["config"]["workload"].split(":")[0], d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["achieved"], r["unit"], r["frac"],
         r["avg_launch_ms"], d["config"].get("timed_region_s", float("nan"))))
print("  multi_gpu:", d["config"].get("multi_gpu"))
for e in d["config"].get("secondary", []):
    print("  secondary:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k != "workload"})
if "secondary_wall_s" in d["config"]:
    print("  secondary wall: %.1f s" % d["config"]["secondary_wall_s"])
print("  cpu_baseline:", d.get("cpu_baseline"))
