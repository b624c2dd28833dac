#!/usr/bin/env python3
"""Top kernels of a rocprofv3 kernel_stats.csv (+ the bench line of a log): tools/show_stats.py <stats.csv> [bench.log]"""
import csv
import json
import sys

for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-70s calls %5s  avg %9.4f ms  %6s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void vsm::", "")[:70], r["Calls"],
                                                    float(r["AverageNs"]) / 1e6, r["Percentage"]))
if len(sys.argv) > 2:
    for l in open(sys.argv[2]):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print("bench: %.0f %s, frac %.4f, avg launch %.3f ms" % (d["value"], d["unit"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]))
