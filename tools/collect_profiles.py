#!/usr/bin/env python3
"""Copy what tools/profile_all.sh left under gpurun_out/prof_<tag>_* into profiles/ (the tracked
evidence) and rebuild profiles/traffic.json (HBM bytes per launch per kernel, read by bench.py).
  python tools/collect_profiles.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}  # merged: a partial re-profile updates its kernels only
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("prof_"):]
    stats = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{name}_kernel_stats.csv"))
    summ = os.path.join(d, "summary.json")
    if os.path.exists(summ) and os.path.getsize(summ) > 0:
        shutil.copy(summ, os.path.join(ROOT, "profiles", f"{name}_summary.json"))
        for k, v in json.load(open(summ))["kernels"].items():
            if "hbm_bytes_per_launch" in v and v.get("pct", 0) > 20:  # the workload's dominant kernel(s)
                traffic[k] = {x: v[x] for x in ("hbm_bytes_per_launch", "hbm_read_bytes", "hbm_write_bytes", "avg_ns")}
                traffic[k]["source"] = f"profiles/{name}_summary.json"
    print(name, "stats" if stats else "-", "summary" if os.path.exists(summ) else "-")
if traffic:
    json.dump(traffic, open(tpath, "w"), indent=1)
    print("traffic.json:", len(traffic), "kernels")
