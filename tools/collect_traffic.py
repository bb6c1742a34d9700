#!/usr/bin/env python3
"""Reads the rocprofv3 --pmc passes made by `bash tools/run_pmc.sh <tag> ...` (FETCH_SIZE, WRITE_SIZE in KiB,
separate runs of tools/spmm_micro.py) and writes profiles/spmm_hbm_traffic.json: measured HBM-side bytes per
launch of the inner pass of the Laplacian chain, with the workload and batch shape it was taken on (bench.py
prints the figure only for that workload).  FETCH_SIZE is doubled: on gfx950 it tallies the 128-byte
requests of wide coalesced reads at 64 bytes.
usage: python tools/collect_traffic.py <tag> [kernel substring]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "spmm_win<2, true, 1>"


def mean_counter(name):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{name}", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / max(len(vals), 1), len(vals)


fetch_kib, nf = mean_counter("FETCH_SIZE")
write_kib, nw = mean_counter("WRITE_SIZE")
log = open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_FETCH_SIZE.log")).read()
mo = re.search(r"^(\w+): m=(\d+) nnz=(\d+) k=(\d+)", log, re.M)
out = {"kernel": kernel, "workload": mo.group(1), "m": int(mo.group(2)), "nnz": int(mo.group(3)), "k": int(mo.group(4)),
       "launches": nf, "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
       "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/spmm_micro.py (one training "
                 "batch of the workload's shape); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts "
                 "wide reads at half their size)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
