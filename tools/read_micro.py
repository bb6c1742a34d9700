import csv, glob, os, sys
for d in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "micro_*"))):
    if not os.path.isdir(d):
        continue
    fs = sorted(glob.glob(d + "/*/*kernel_stats.csv"), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        if any(t in r["Name"] for t in sys.argv[1:] or ["spmm_win", "spmm_rows<32", "lap_scale", "lap_colsum", "loss_finish"]):
            print(os.path.basename(d), f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:.2f}")
