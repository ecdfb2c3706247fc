#!/usr/bin/env python3
"""profiles/blend_traffic.json from the rocprofv3 --pmc passes of scripts/gpu_pmc.sh (gpurun_out/pmc/*).

k_blend's HBM traffic and VALU instruction count per launch, for the two configurations bench.py runs: the timed region
(several frames in flight: 6 workgroups per CU) and the one-frame leg (7 per CU), told apart by the dispatch's grid size.
The file is stamped with the build id of the library the counters were collected on (bench.py refuses other builds).
gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide read -> doubled (MI355X_MICROARCH.md, HBM); both counters
are in KB."""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
build_id = sys.argv[2] if len(sys.argv) > 2 else open(os.path.join(src, "build_id.txt")).read().strip()

vals = {}   # (grid, counter) -> [values]
dur = {}
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_blend" not in r["Kernel_Name"]:
            continue
        grid = int(r["Grid_Size"]) // 256
        vals.setdefault((grid, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        dur.setdefault(grid, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
grids = sorted({g for g, _ in vals})
if len(grids) < 2:
    raise SystemExit("expected k_blend dispatches of two grid sizes (one-frame leg and frames-in-flight leg), found %s" % grids)


def leg(grid, what):
    med = lambda c: statistics.median(vals[(grid, c)]) if (grid, c) in vals else None
    fetch, write = med("FETCH_SIZE"), med("WRITE_SIZE")
    out = {"config": what, "workgroups": grid, "dispatches": len(vals.get((grid, "FETCH_SIZE"), [])),
           "fetch_size_kb_raw": fetch, "write_size_kb_raw": write,
           "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
           "valu_wave_instructions_per_launch": med("SQ_INSTS_VALU"),
           "salu_wave_instructions_per_launch": med("SQ_INSTS_SALU"),
           "lds_wave_instructions_per_launch": med("SQ_INSTS_LDS"),
           "serialised_duration_us": statistics.median(dur[grid])}
    w, a, i = med("SQ_WAVE_CYCLES"), med("SQ_WAIT_ANY"), med("SQ_WAIT_INST_ANY")
    if w and a:
        out["sq_wait_any_over_wave_cycles"] = a / w
    if w and i:
        out["sq_wait_inst_any_over_wave_cycles"] = i / w
    c, t = med("SQ_LDS_BANK_CONFLICT"), med("SQ_LDS_IDX_ACTIVE")
    if c is not None and t:
        out["lds_bank_conflict_over_idx_active"] = c / t
    return out


doc = {
    "kernel": "k_blend",
    "build_id": build_id,
    "workload": "C3 exact mode, bench.py --steps 12 --warmup 4 under rocprofv3 --pmc (kernels serialised by the profiler); medians per configuration",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, separate passes (scripts/gpu_pmc.sh); summary of all kernels in profiles/r02_pmc_c3_exact.txt",
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; rocprofv3 reports both in KB (x1024); for this kernel's 32-byte record gathers the doubling is an upper bound: true traffic lies between fetch_raw + write and 2 x fetch_raw + write",
    "frames_in_flight": leg(grids[0], "GSR_FLAG_THROUGHPUT: %d workgroups (the timed region of the default bench)" % grids[0]),
    "one_frame": leg(grids[-1], "default context: %d workgroups (bench.py's one_frame_in_flight leg)" % grids[-1]),
    "note": "counts L2<->fabric traffic incl. Infinity Cache hits; the 32 MB record array is cache resident. Above the algorithmic 32D+16P because each 32-B record gather pulls a whole line and the segment partials are written, then re-read by the fold at the end of k_blend.",
}
json.dump(doc, open(os.path.join(ROOT, "profiles", "blend_traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
