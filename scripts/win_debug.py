#!/usr/bin/env python3
"""Diagnostic: the front window of heavy bins on one C3 pose -- work-item counts and image differences per cut."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import numpy as np
import gsplat_hip as gh
cfg = gh.synth.CONFIGS["C3"]
W, H = cfg["width"], cfg["height"]
scene = gh.Scene(); scene.setData(gh.synth.config_rows("C3"))
cam = gh.orbit_camera(int(sys.argv[1]) if len(sys.argv) > 1 else 5, 120, W, H, cfg["fx"])
os.environ["GSR_LONG_ITEMS"] = "1"
ref = None
for env in ({"GSR_WIN_FROM": "0"}, {}, {"GSR_WIN_FROM": "600", "GSR_WIN_LEN": "512", "GSR_WIN_SEGS": "2"}):
    os.environ.update(env)
    r = gh.HIPRenderer(W, H)
    for k in env: del os.environ[k]
    r.render(scene, cam)
    img = r.readPixelsFloat()
    bt = r.bin_totals()
    print(env, r.work_items(), "bins >= 2048:", int((bt >= 2048).sum()), "max bin", int(bt.max()),
          "diff vs first:", 0.0 if ref is None else float(np.abs(img - ref).max()))
    if ref is None: ref = img
    r.dispose()
