#!/usr/bin/env python3
"""How far is the browser's canvas from the fp32 image parity is defined on?  The reference blends into an RGBA8
drawing buffer, re-quantising after every splat (WebGLRenderer.ts:38,139-142,282-285).  The oracle's mode 2 models
that ROP; this script prints max / mean |mode 2 - mode 0| per configuration (CPU only; numbers quoted in DESIGN.md)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh  # noqa: E402
from oracle import oracle as O  # noqa: E402

for name, poses in (("C1", (0, 40, 80)), ("C2", (13,))):
    cfg = gh.synth.CONFIGS[name]
    data, pos = O.scene_pack(gh.synth.config_rows(name))
    for k in poses:
        cam = gh.orbit_camera(k, width=cfg["width"], height=cfg["height"], fx=cfg["fx"])
        v, p, vp = cam.f32()
        di, _, _ = O.sort(vp, pos)
        rec, bbox, raw = O.project(data, v, p, cam.fx, cam.fy, cfg["width"], cfg["height"])
        a = O.render(di, raw, rec, bbox, cfg["width"], cfg["height"], 0).astype(np.float64)
        b = O.render(di, raw, rec, bbox, cfg["width"], cfg["height"], 2).astype(np.float64)
        d = np.abs(a - b)
        q = np.abs(np.round(np.clip(a, 0, 1) * 255) / 255 - b)   # against the fp32 image quantised once at the end
        print("%s pose %3d: |rop8 - fp32| max %.4f (%.1f LSB) mean %.5f (%.2f LSB); vs fp32 rounded once: max %.1f LSB mean %.2f LSB; "
              "pixels off by >= 2 LSB: %.1f %%" % (name, k, d.max(), d.max() * 255, d.mean(), d.mean() * 255, q.max() * 255, q.mean() * 255,
                                                 100.0 * (q.max(axis=2) * 255 >= 1.5).mean()))
