#!/usr/bin/env python3
"""Render a few frames of a large synthetic scene (default 20 M splats, 1080p) for kernel traces of the streaming
kernels outside the launch-latency regime: python scripts/big_scene_frames.py [n] [frames] [width] [height]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))
import gsplat_hip as gh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
rows = gh.synth.synth_rows(n, 77, sigma=1.5, s_lo=0.002, s_hi=0.02)
r = gh.HIPRenderer(W, H, timing=True)
r.set_scene_rows(rows)
del rows
for k in range(frames):
    r.set_camera(gh.orbit_camera(33 + 7 * k, width=W, height=H, fx=1132.0 * W / 1920))
    r.render_async(); r.sync()
    st = r.stats()
    print("frame %d: %.2f ms (project %.3f sort %.3f bin %.3f blend %.3f)  V=%d bin_entries=%d" % (
        k, st["ms_total"], st["ms_project_key"], st["ms_sort"], st["ms_bin"], st["ms_blend"], st["visible"], st["bin_entries"]), flush=True)
r.dispose()
