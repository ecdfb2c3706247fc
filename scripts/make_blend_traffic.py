#!/usr/bin/env python3
"""profiles/blend_traffic.json from the rocprofv3 --pmc passes of `scripts/gpu.sh pmc solo` / `pmc inflight` (gpurun_out/pmc/*).

The compositor's HBM traffic and VALU instruction count per launch, for the two configurations bench.py runs: the timed
region (several frames in flight on throughput contexts: k_blend, one wave per tile) and the one-frame leg (k_blend2, two
waves per tile).  Each is profiled in runs of its own (bench.py --timed-only with three frames in flight, and with one):
every compositor dispatch of a run belongs to one configuration.
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

vals = {}   # (leg, counter) -> [values]; leg 0 = pmc/inflight_*, leg 1 = pmc/solo_* (scripts/gpu.sh pmc: one configuration per run)
dur = {}
grid_of = {}
for which, legname in ((0, "inflight"), (1, "solo")):
    for f in glob.glob(os.path.join(src, legname + "_*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_blend" not in r["Kernel_Name"]:
                continue
            vals.setdefault((which, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
            dur.setdefault(which, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            grid_of[which] = int(r["Grid_Size"]) // max(1, int(r.get("Workgroup_Size") or 256))
grids = [0, 1]
if not all((g, "FETCH_SIZE") in vals for g in grids):
    raise SystemExit("expected k_blend dispatches of both legs (inflight_*, solo_*) in %s" % src)


def leg(grid, what):
    med = lambda c: statistics.median(vals[(grid, c)]) if (grid, c) in vals else None
    fetch, write = med("FETCH_SIZE"), med("WRITE_SIZE")
    out = {"config": what, "workgroups": grid_of[grid], "dispatches": len(vals.get((grid, "FETCH_SIZE"), [])),
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
    "kernel": "k_blend (frames_in_flight) / k_blend2 (one_frame)",
    "build_id": build_id,
    "workload": "C3 exact mode, bench.py --timed-only --steps 24 --warmup 4 under rocprofv3 --pmc (kernels serialised by the profiler); medians per configuration",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, separate passes (scripts/gpu.sh pmc); summary of all kernels in profiles/r04_pmc_c3.txt",
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; rocprofv3 reports both in KB (x1024); for this kernel's 32-byte record gathers the doubling is an upper bound: true traffic lies between fetch_raw + write and 2 x fetch_raw + write",
    "frames_in_flight": leg(0, "GSR_FLAG_THROUGHPUT contexts, three frames in flight (the timed region of the default bench)"),
    "one_frame": leg(1, "exact context, one frame at a time (bench.py --frames-in-flight 1; the one_frame_in_flight leg uses the same)"),
    "note": "counts L2<->fabric traffic incl. Infinity Cache hits; the 32 MB record array is cache resident. Above the algorithmic 32D+16P because each 32-B record gather pulls a whole line and the segment partials are written, then re-read by the fold at the end of k_blend.",
}
json.dump(doc, open(os.path.join(ROOT, "profiles", "blend_traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
