#!/usr/bin/env python3
import csv, glob, json, os, sys
for f in sorted(glob.glob("gpurun_out/bench1*.json")):
    try:
        d = json.load(open(f))
        print("%-34s fps %7.1f ms %.3f" % (os.path.basename(f), d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["stage_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
fs = sorted(glob.glob("gpurun_out/prof1/*/*kernel_stats.csv"), key=os.path.getmtime)
if fs:
    print(fs[-1])
    for r in csv.DictReader(open(fs[-1])):
        print("  %-36s calls %4s avg %8.1f us %6s%%" % (r["Name"].split("(")[0].replace("void ", "").replace("gsr::", "")[:36], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
