#!/usr/bin/env python3
"""Phase times inside the radix scatters and the bin scatter (diagnostic build: scripts/build_exp.sh kstamps "-DGSR_KSTAMPS")."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "kstamps"
config = sys.argv[2] if len(sys.argv) > 2 else "C3"
lib = os.path.join(ROOT, "gsplat.js_amd", "lib_exp", name, "libgsplat_hip.so")
cfg = gh.synth.CONFIGS[config]
W, H = cfg["width"], cfg["height"]
scene = gh.Scene()
scene.setData(gh.synth.config_rows(config))
r = gh.HIPRenderer(W, H, timing=True, lib_path=lib)
for k in range(3):
    r.render(scene, gh.orbit_camera(11 * k, 120, W, H, cfg["fx"]))
L = r._L


def show(title, a, names):
    a0 = a
    a = a[a[:, 5] != 0].astype(np.int64)
    t0 = a[:, 0].min()
    print("%s: %d workgroups, kernel span %.1f us (first start -> last end)" % (title, len(a), 0.01 * (a[:, 5].max() - t0)))
    print("   workgroup start after the first: p50 %.1f p90 %.1f max %.1f us" % tuple(0.01 * np.percentile(a[:, 0] - t0, q) for q in (50, 90, 100)))
    for k, nm in enumerate(names):
        d = (a[:, k + 1] - a[:, k]) * 0.01
        print("   %-34s mean %5.2f  p90 %5.2f  max %5.2f us" % (nm, d.mean(), np.percentile(d, 90), d.max()))
    life = (a[:, 5] - a[:, 0]) * 0.01
    print("   workgroup lifetime                 mean %5.2f  p90 %5.2f  max %5.2f us" % (life.mean(), np.percentile(life, 90), life.max()))
    ids = np.nonzero(a0[:, 5] != 0)[0]
    order = np.argsort(-life)[:8]
    print("   slowest workgroups (blockIdx: lifetime us): " + ", ".join("%d: %.1f" % (ids[k], life[k]) for k in order))
    print("   lifetime by position: first 4 " + " ".join("%.1f" % x for x in life[:4]) + " | p99 %.1f | last 4 " % np.percentile(life, 99) + " ".join("%.1f" % x for x in life[-4:]))


buf = np.zeros(2 * 4096 * 8, dtype=np.uint32)
L.gsr_debug_sort_stamps(ctypes.c_void_p(buf.ctypes.data))
names = ["zero LDS + barrier", "key loads, LDS count, totals scan", "digit starts + barrier", "phase 2 (per-wave offsets)", "phase 3 (rank + scatter)"]
show("k_scatter pass 1 (8 bits)", buf.reshape(2, 4096, 8)[0], names)
show("k_scatter pass 2 (9 bits)", buf.reshape(2, 4096, 8)[1], names)
b2 = np.zeros(4096 * 8, dtype=np.uint32)
L.gsr_debug_bin_stamps(ctypes.c_void_p(b2.ctypes.data))
show("k_bin_scatter", b2.reshape(4096, 8), ["zero LDS + barrier", "idx/rect loads + base table loads", "phase 1 (lane sets, counts)", "phase 2 (group offsets)", "phase 3 (slots + scatter)"])
r.dispose()
