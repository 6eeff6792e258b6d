#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by profiles/collect.sh into the small summaries kept under profiles/:
   <tag>_kernel_stats.csv / <tag>_kernel_stats_serial.csv (copies of rocprofv3's kernel_stats.csv),
   <tag>_pmc_traffic_serial.json (FETCH_SIZE / WRITE_SIZE in KB per launch and per bench step),
   <tag>_pmc_sq.json (SQ / TCP counters per launch).
Usage: summarize.py <gpurun_out dir> <tag> [steps-in-the-pmc-runs (default 3 = 2 timed + 1 warm-up)]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
nsteps = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0 + 2.0  # warm-up + timed + the 2 profiling steps of bench.py


def kname(full):
    m = re.search(r"(k_[a-z0-9_]+)", full)
    if not m:
        return None
    k = m.group(1)
    t = re.search(r"k_knn_cov_far<(\d+)", full)
    if t:
        k = "k_knn_cov_far" if t.group(1) == "16" else "k_knn_cov_far_groups"
    return k


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k].add(r["Dispatch_Id"])
    return acc, n


for suffix in ("stats", "stats_serial"):
    for f in glob.glob(os.path.join(out, f"{tag}_{suffix}", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_kernel_{suffix}.csv"))

traffic = {"_batch_pairs": int(os.environ.get("GFS_BENCH_BATCH", "512"))}  # bench.py's default --batch (the PMC passes use it)
passes_with_rows = 0
for c, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc, n = counters(os.path.join(out, f"{tag}_pmc_{c}"))
    passes_with_rows += bool(acc)
    for k, v in acc.items():
        t = traffic.setdefault(k, {})
        launches = len(n[k])
        t[f"{key}_kb_per_launch"] = round(v[c] / launches, 1)
        t["launches_per_step"] = round(launches / nsteps, 2)
        t[f"{key}_kb_per_step"] = round(v[c] / nsteps, 1)
# a traffic file is only worth keeping when BOTH passes produced rows for every kernel in it (round 4 committed one whose WRITE_SIZE pass
# had been killed; bench.py read it and died): incomplete kernels are dropped, an incomplete collection writes nothing
incomplete = [k for k, t in traffic.items() if not k.startswith("_") and not ("fetch_kb_per_step" in t and "write_kb_per_step" in t)]
for k in incomplete:
    del traffic[k]
tpath = os.path.join(out, f"{tag}_pmc_traffic_serial.json")
if passes_with_rows == 2 and len(traffic) > 1:
    json.dump(traffic, open(tpath, "w"), indent=1)
    if incomplete:
        print("summarize.py: kernels seen by only one PMC pass, dropped:", incomplete, file=sys.stderr)
else:
    if os.path.exists(tpath):
        os.remove(tpath)
    print(f"summarize.py: NOT writing {os.path.basename(tpath)}: {passes_with_rows} of the 2 PMC passes produced rows", file=sys.stderr)

sq = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
    acc, n = counters(os.path.join(out, f"{tag}_{d}"))
    for k, v in acc.items():
        sq.setdefault(k, {}).update({a: round(b / len(n[k])) for a, b in v.items()})
        sq[k]["launches"] = len(n[k])
json.dump(sq, open(os.path.join(out, f"{tag}_pmc_sq.json"), "w"), indent=1)
print("summaries written:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(out, f"{tag}_*.json")) + glob.glob(os.path.join(out, f"{tag}_kernel_*.csv"))))
