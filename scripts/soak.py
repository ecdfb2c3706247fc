#!/usr/bin/env python3
"""Soak: thousands of frames through three throughput contexts and one exact context, several scenes, progress lines.
Run under a timeout; a stall shows as a missing progress line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name in ("C2", "C3", "C1", "C4"):
    cfg = gh.synth.CONFIGS[name]; W, H = cfg["width"], cfg["height"]
    scene = gh.Scene(); scene.setData(gh.synth.config_rows(name))
    poses = [gh.orbit_camera(k, 120, W, H, cfg["fx"]).f32() for k in range(120)]
    # three throughput contexts, one exact context, and two band contexts (ranks of a multi-GPU frame: survivors packed,
    # gathered and sorted alone) -- all with frames in flight at once
    bw = (W // 4 // 32) * 32
    rs = [gh.HIPRenderer(W, H, throughput=True) for _ in range(3)] + [gh.HIPRenderer(W, H)] + \
         [gh.HIPRenderer(W, H, throughput=True, band=(bw, 2 * bw)), gh.HIPRenderer(W, H, throughput=True, band=(3 * bw, W))]
    for r in rs:
        r.render(scene, gh.orbit_camera(0, 120, W, H, cfg["fx"]))
    t0 = time.perf_counter()
    for k in range(frames if name != "C4" else frames // 6):
        r = rs[k % len(rs)]
        r.set_camera_arrays(*poses[(k * 7) % 120], cfg["fx"], cfg["fx"])
        r.render_async()
        if k % 97 == 96:
            r.sort(gh.orbit_camera(k % 120, 120, W, H, cfg["fx"]))     # a sort-only frame between the rendered ones
        if k % 500 == 499:
            for x in rs: x.sync()
            print("%s: %d frames, %.1f s" % (name, k + 1, time.perf_counter() - t0), flush=True)
    for x in rs: x.sync()
    st = rs[0].stats()
    assert st["overflow_frames"] == 0, st
    for x in rs: x.dispose()
print("soak ok")
