#!/usr/bin/env python3
"""rocprofv3 evidence for one command, condensed into one JSON (copied to profiles/rNN/ by hand).

  python tools/profile_any.py --out gpurun_out/prof_<tag> [--dtype f64|f32] -- <command ...>

Passes (each its own run, as MI355X_MICROARCH.md prescribes -- counters never share a run with --stats):
  stats : --kernel-trace --stats                     per-kernel calls / average duration
  fetch : --kernel-trace --pmc FETCH_SIZE            HBM read  (KiB; doubled: gfx950 tallies 128-B requests at 64 B)
  write : --kernel-trace --pmc WRITE_SIZE            HBM write (KiB, uncalibrated, taken as is)
  sq1   : wave / wait / MFMA-busy / LDS cycles
  sq2   : instruction counts incl. MFMA MOPS of the dtype
Output: <out>/summary.json  {command, kernels: {name: {calls, avg_ms, pct, fetch_bytes_per_launch, write_bytes_per_launch,
        sq: {counter: sum over launches}, launches_in_pmc_pass}}}  and <out>/kernel_stats.csv.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)


def counters(d):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--skip-pmc", action="store_true")
    ap.add_argument("--points-per-run", type=int, default=0,
                    help="spectral points of one run of the command: adds hbm_bytes_per_point_whole_run (all vsm:: kernels)")
    ap.add_argument("--runs", type=int, default=1, help="runs of the workload inside the command (warm-up + timed)")
    ap.add_argument("--sources", default="", help="comma-separated source files / globs (relative to the repo root) that decide what "
                    "this tag measures: their SHA-256 goes into summary.json as `tag_sources_hash`")
    ap.add_argument("--skip-if-unchanged", default="", help="path of a committed summary.json of the same tag: when its "
                    "tag_sources_hash equals the current one the tag is not re-profiled (a Raman edit does not re-take C2)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tag_hash = None
    if a.sources:
        import hashlib
        h = hashlib.sha256()
        files = sorted({f for pat in a.sources.split(",") for f in glob.glob(os.path.join(root, pat.strip()))})
        for f in files:
            h.update(os.path.relpath(f, root).encode() + b"\0" + open(f, "rb").read())
        h.update(" ".join(cmd).encode())
        tag_hash = h.hexdigest()[:16]
        if a.skip_if_unchanged and os.path.exists(a.skip_if_unchanged):
            try:
                if json.load(open(a.skip_if_unchanged)).get("tag_sources_hash") == tag_hash:
                    print("tag unchanged (tag_sources_hash %s = %s): not re-profiled" % (tag_hash, a.skip_if_unchanged), flush=True)
                    return
            except (OSError, ValueError):
                pass
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    mops = "SQ_INSTS_VALU_MFMA_MOPS_F64" if a.dtype == "f64" else "SQ_INSTS_VALU_MFMA_MOPS_F32"
    passes = {"stats": ["--kernel-trace", "--stats"]}
    if not a.skip_pmc:
        passes.update({
            "fetch": ["--kernel-trace", "--pmc", "FETCH_SIZE"],
            "write": ["--kernel-trace", "--pmc", "WRITE_SIZE"],
            "sq1": ["--kernel-trace", "--pmc", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                    "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
            "sq2": ["--kernel-trace", "--pmc", mops, "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                    "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM"],
        })
    for name, flags in passes.items():
        d = os.path.join(os.path.abspath(a.out), name)
        with open(os.path.join(a.out, name + ".log"), "w") as log:
            rc = subprocess.call(["rocprofv3", "--output-format", "csv"] + flags + ["-d", d, "-o", "p", "--"] + cmd, stdout=log,
                                 stderr=subprocess.STDOUT, env=env, cwd=os.getcwd())
        print("pass %-5s rc=%d" % (name, rc), flush=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vsmartmom_jl_amd import _lib as vsm_lib      # (identity of the library the profiled command loads: csrc/Makefile)
    summary = {"command": " ".join(cmd), "dtype": a.dtype, "library": vsm_lib.build_info(), "tag_sources_hash": tag_hash, "tag_sources": a.sources,
               "corrections": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                              "requests at 64 B); WRITE_SIZE uncalibrated, taken as is", "kernels": {}}
    for f in glob.glob(os.path.join(a.out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(a.out, "kernel_stats.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
                if short(r["Name"]).startswith("vsm::"):
                    summary["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                            "pct": float(r["Percentage"])}
    if not a.skip_pmc:
        fa, fl = counters(os.path.join(a.out, "fetch"))
        wa, wl = counters(os.path.join(a.out, "write"))
        sq = {}
        for sub in ("sq1", "sq2"):
            ag, ln = counters(os.path.join(a.out, sub))
            for k, v in ag.items():
                sq.setdefault(k, {}).update(v)
                sq[k]["launches"] = ln[k]
        for k in set(fa) | set(sq):
            if not k.startswith("vsm::"):
                continue
            rec = summary["kernels"].setdefault(k, {})
            if k in fa:
                rec["fetch_bytes_per_launch"] = 2.0 * 1024.0 * fa[k]["FETCH_SIZE"] / fl[k]
            if k in wa:
                rec["write_bytes_per_launch"] = 1024.0 * wa[k]["WRITE_SIZE"] / wl[k]
            if k in sq:
                rec["sq_sum_over_launches"] = sq[k]
                s = sq[k]
                if s.get("SQ_WAVE_CYCLES"):
                    rec["wait_any_frac"] = s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"]
                    rec["wait_inst_any_frac"] = s.get("SQ_WAIT_INST_ANY", 0) / s["SQ_WAVE_CYCLES"]
                if s.get("SQ_BUSY_CYCLES"):
                    rec["mfma_busy_over_sq_busy"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / s["SQ_BUSY_CYCLES"]
                rec["mfma_mops_per_launch"] = s.get(mops, 0) / max(s.get("launches", 1), 1)
    if a.points_per_run and not a.skip_pmc:
        tot = sum((v.get("fetch_bytes_per_launch", 0.0) + v.get("write_bytes_per_launch", 0.0)) * v.get("calls", 0)
                  for v in summary["kernels"].values())
        summary["points_per_run"], summary["runs"] = a.points_per_run, a.runs
        summary["hbm_bytes_per_point_whole_run"] = tot / (a.runs * a.points_per_run)
        summary["note"] = ("whole-run HBM bytes per point = sum over the vsm:: kernels of (fetch + write) per launch x calls / "
                           "(runs x points_per_run)")
    json.dump(summary, open(os.path.join(a.out, "summary.json"), "w"), indent=1)
    import shutil
    for name in passes:      # the raw per-dispatch CSVs are tens of MB; only the condensed files travel back
        shutil.rmtree(os.path.join(a.out, name), ignore_errors=True)
    top = sorted(summary["kernels"].items(), key=lambda kv: -kv[1].get("pct", 0))[:8]
    for k, v in top:
        print("%-60s calls=%s avg_ms=%s pct=%s fetch=%s write=%s" % (k[:60], v.get("calls"), v.get("avg_ms"), v.get("pct"),
                                                                     v.get("fetch_bytes_per_launch"), v.get("write_bytes_per_launch")))


if __name__ == "__main__":
    main()
