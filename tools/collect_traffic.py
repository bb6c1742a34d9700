#!/usr/bin/env python3
"""Reads the two rocprofv3 --pmc passes made by tools/run_pmc.sh (FETCH_SIZE, WRITE_SIZE, in KiB)
and writes profiles/spmm_hbm_traffic.json: measured HBM-side bytes per launch of the inner pass
of the Laplacian chain.  FETCH_SIZE is doubled: on gfx950 it tallies 128-byte requests of wide
coalesced reads at 64 bytes."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(pattern, kernel_substr, counter):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / max(len(vals), 1), len(vals)


kernel = sys.argv[1] if len(sys.argv) > 1 else "spmm_tiled<true, 1>"
fetch_kib, nf = mean_counter("pmc_fetch", kernel, "FETCH_SIZE")
write_kib, nw = mean_counter("pmc_write", kernel, "WRITE_SIZE")
out = {"kernel": kernel, "launches": nf, "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
       "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/spmm_micro.py (one "
                 "Amazon2M-shape batch, k=47); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE "
                 "counts wide reads at half their size)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
