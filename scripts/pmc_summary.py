#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, median counter value per dispatch.
  python scripts/pmc_summary.py [gpurun_out/pmc] [leg]     leg = solo | inflight (scripts/gpu.sh pmc), default solo"""
import csv
import glob
import sys
from collections import defaultdict

def main(root, leg="solo"):
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in sorted(set(glob.glob(root + "/" + leg + "_*/*/*counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsr::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(agg, key=lambda k: -sum(dur[k])):
        d = sorted(dur[k])
        print("%-28s n=%4d  dur_us median %.1f" % (k[:28], len(d), d[len(d) // 2]))
        for c in sorted(agg[k]):
            v = sorted(agg[k][c])
            print("     %-24s %16.1f" % (c, v[len(v) // 2]))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc", sys.argv[2] if len(sys.argv) > 2 else "solo")
