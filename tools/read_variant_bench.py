#!/usr/bin/env python3
"""ms per step and the per-step kernels (at least 35 calls) of the runs made by tools/run_variant_bench.sh.
usage: python tools/read_variant_bench.py base [name ...] [-- substring ...]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
subs = args[args.index("--") + 1:] if "--" in args else ["gram_", "orth_", "narrow_fwd", "spmm_win", "lap_scale"]
names = args[:args.index("--")] if "--" in args else args
for v in names or ["base"]:
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"bp_{v}", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
    d = json.loads(open(os.path.join(ROOT, "gpurun_out", f"bp_{v}.json")).read().strip().split("\n")[-1])
    print(v, f"{d['ms_per_step']:.4f} ms/step")
    for r in csv.DictReader(open(f)):
        if any(t in r["Name"] for t in subs) and int(r["Calls"]) >= 35:
            print(f"   {r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:8.2f}")
