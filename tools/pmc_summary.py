#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per launch and kernel.
usage: pmc_summary.py <dir with *counter_collection.csv> [name filter ...]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
filters = sys.argv[2:]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if filters and not any(x in name for x in filters):
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        key = (short, r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
print(f"{'kernel':72} {'counter':14} {'launches':>8} {'mean/launch':>16}")
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k:72} {c:14} {n:8d} {s / n:16.1f}")
