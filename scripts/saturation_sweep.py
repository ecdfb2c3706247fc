#!/usr/bin/env python3
"""Exhaustive check of the saturation skip over a whole orbit: for every pose the image with the skip equals the image
without it bit for bit (f32 and RGBA8), with long and with short work items, on the configs given (default C3 C2 C4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import numpy as np
import gsplat_hip as gh

bad = 0
for name in (sys.argv[1:] or ["C3", "C2", "C4"]):
    cfg = gh.synth.CONFIGS[name]; W, H = cfg["width"], cfg["height"]
    scene = gh.Scene(); scene.setData(gh.synth.config_rows(name))
    step = 1 if name != "C4" else 6
    for pol in ("1", "0"):
        os.environ["GSR_LONG_ITEMS"] = pol
        os.environ["GSR_SATURATE"] = "0"
        a = gh.HIPRenderer(W, H)
        del os.environ["GSR_SATURATE"]
        b = gh.HIPRenderer(W, H)
        n = 0
        for k in range(0, 120, step):
            cam = gh.orbit_camera(k, 120, W, H, cfg["fx"])
            a.render(scene, cam); b.render(scene, cam)
            same = np.array_equal(a.readPixelsFloat(), b.readPixelsFloat()) and np.array_equal(a.readPixels(), b.readPixels())
            n += 1
            if not same:
                bad += 1
                print("MISMATCH", name, "long" if pol == "1" else "short", "pose", k, flush=True)
        a.dispose(); b.dispose()
        print("%s %s items: %d poses compared" % (name, "long" if pol == "1" else "short", n), flush=True)
    del os.environ["GSR_LONG_ITEMS"]
print("mismatches:", bad)
sys.exit(1 if bad else 0)
