#!/usr/bin/env python3
"""Averages per kernel of the rocprofv3 --pmc passes made by tools/run_pmc.sh.
usage: python tools/read_pmc.py <tag> [kernel substring ...]"""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
subs = sys.argv[2:] or ["spmm_win", "spmm_rows<32"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in subs):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
