"""Timeline analysis of a rocprofv3 --kernel-trace CSV (scripts/gpu.sh trace NAME ...): what the chip runs at every instant
of the steady state with several frames in flight -- compositor only, front-end kernels only, both, nothing -- and per
kernel the average duration, so that `frames take turns` vs `frames overlap` is a measured statement.
usage: python scripts/trace_timeline.py gpurun_out/trace_NAME/*/*_kernel_trace.csv [skip_fraction]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("gsr::") or "k_" in r["Kernel_Name"]]
rows = [r for r in rows if "rocclr" not in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n.replace("gsr::", "").split("(")[0]
    r["n"] = n
rows.sort(key=lambda r: r["s"])
t_first, t_last = rows[0]["s"], max(r["e"] for r in rows)
t0 = t_first + (t_last - t_first) * skip
t1 = t_last - (t_last - t_first) * 0.05
sel = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
is_comp = lambda n: n.startswith("k_blend")
ev = []
for r in sel:
    c = is_comp(r["n"])
    ev.append((r["s"], 1, c))
    ev.append((r["e"], -1, c))
ev.sort()
nc = nf = 0
last = ev[0][0]
acc = defaultdict(int)
for t, d, c in ev:
    key = ("comp%d" % min(nc, 3)) + ("+front%d" % min(nf, 3))
    acc[key] += t - last
    last = t
    if c: nc += d
    else: nf += d
span = ev[-1][0] - ev[0][0]
frames = sum(1 for r in sel if is_comp(r["n"]))
print("window %.3f ms, %d compositor launches -> %.1f us per frame" % (span / 1e6, frames, span / 1e3 / max(frames, 1)))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("  %-16s %6.1f %%   %7.1f us per frame" % (k, 100.0 * acc[k] / span, acc[k] / 1e3 / max(frames, 1)))
dur = defaultdict(list)
for r in sel:
    dur[r["n"]].append(r["e"] - r["s"])
print("kernel averages (us), launches:")
tot = 0
for n in sorted(dur, key=lambda n: -sum(dur[n])):
    a = sum(dur[n]) / len(dur[n]) / 1e3
    per_frame = sum(dur[n]) / 1e3 / max(frames, 1)
    tot += per_frame
    print("  %-60s %8.1f  x%-4d  %7.1f per frame" % (n[:60], a, len(dur[n]), per_frame))
print("  sum of kernel durations per frame: %.1f us" % tot)
# queue waits: time between a kernel's start and the end of the previous kernel of the same stream (same Queue_Id)
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
gaps = defaultdict(list)
for q, rs in byq.items():
    rs.sort(key=lambda r: r["s"])
    for a, b in zip(rs[:-1], rs[1:]):
        gaps[b["n"]].append(b["s"] - a["e"])
print("gap in front of a kernel on its own queue (us, average):")
for n in sorted(gaps, key=lambda n: -sum(gaps[n])):
    print("  %-60s %8.1f" % (n[:60], sum(gaps[n]) / len(gaps[n]) / 1e3))
