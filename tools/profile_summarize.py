#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_round.sh into small JSON/CSV summaries (copied to profiles/)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
POINTS = 4096


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name


def counters(sub):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


summary = {"workload": "C2", "N": 60, "dtype": "f64", "points_per_launch": POINTS, "command": "python bench.py --points 4096 --steps 1 --warmup 0 --no-cpu-baseline",
           "corrections": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                          "requests at 64 B); WRITE_SIZE uncalibrated, taken as is", "kernels": {}}
fa, fl = counters("fetch")
wa, wl = counters("write")
for k in fa:
    if not k.startswith("vsm::"):
        continue
    n = fl[k]
    fetch = 2.0 * 1024.0 * fa[k]["FETCH_SIZE"] / n
    write = 1024.0 * wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wl.get(k, 1), 1)
    summary["kernels"][k] = {"launches": n, "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
                             "hbm_bytes_per_point": (fetch + write) / POINTS}
sq = {}
for sub in ("sq1", "sq2"):
    a, l = counters(sub)
    for k, v in a.items():
        if k.startswith("vsm::"):
            sq.setdefault(k, {"launches": l[k]}).update(v)
summary["sq_counters_sum_over_launches"] = sq
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
# kernel stats of the default bench command
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
print(json.dumps({k: {kk: vv for kk, vv in v.items()} for k, v in summary["kernels"].items()}, indent=1))
for r in open(os.path.join(out, "kernel_stats.csv")).read().splitlines()[:8]:
    print(r)
