#!/usr/bin/env python3
"""Create / render / resize / destroy contexts in a loop and watch the device's free memory (hipMemGetInfo via torch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))
import torch
import gsplat_hip as gh
cfg = gh.synth.CONFIGS["C1"]
scene = gh.Scene(); scene.setData(gh.synth.config_rows("C1"))
free = []
for it in range(120):
    r = gh.HIPRenderer(cfg["width"], cfg["height"], timing=(it % 2 == 0), throughput=(it % 3 == 0), band=(0, 320) if it % 5 == 0 else None)
    for k in range(6):
        r.render(scene, gh.orbit_camera(k, width=r.width, height=r.height, fx=cfg["fx"]))
        if k == 3:
            r.setSize(512 + 32 * (it % 4), 384)
    r.lastDepthIndex()
    if it % 6 == 0:    # a 4K framebuffer: the two-level binning's buffers come and go with the context too
        r.setSize(3840, 2160)
        r.render(scene, gh.orbit_camera(1, width=3840, height=2160, fx=2.0 * cfg["fx"]))
    if it % 4 == 0:    # sort-only frames (their own frame slots)
        r.sort(gh.orbit_camera(2, width=r.width, height=r.height, fx=cfg["fx"]))
        r.sort(gh.orbit_camera(3, width=r.width, height=r.height, fx=cfg["fx"]))
    r.dispose()
    if it % 20 == 19:
        free.append(torch.cuda.mem_get_info()[0])
print("free bytes every 20 contexts:", free)
assert max(free) - min(free[1:]) < 64 << 20, "device memory keeps shrinking"
print("ok: no growth")
