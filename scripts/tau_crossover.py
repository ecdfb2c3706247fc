#!/usr/bin/env python3
"""Where do long work items start to pay?  Scenes of the C3 / C2 generators at several sizes (optical depth tau scales with
the splat count), short vs long items (GSR_LONG_ITEMS pinned), one frame at a time and three in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import numpy as np
import gsplat_hip as gh

W, H, fx = 1920, 1080, 1132.0
poses = [gh.orbit_camera(k, 120, W, H, fx).f32() for k in range(120)]
for gen, sizes in (("C3", (250_000, 400_000, 550_000, 700_000)), ("C2", (300_000, 600_000, 1_000_000, 1_600_000))):
    c = gh.synth.CONFIGS[gen]
    for n in sizes:
        rows = gh.synth.synth_rows(n, c["seed"], c["sigma"], c["s_lo"], c["s_hi"])
        scene = gh.Scene(); scene.setData(rows)
        out = []
        for pol in ("0", "1"):
            os.environ["GSR_LONG_ITEMS"] = pol
            for F in (1, 3):
                rs = [gh.HIPRenderer(W, H, throughput=F > 1) for _ in range(F)]
                for r in rs:
                    r.render(scene, gh.orbit_camera(0, 120, W, H, fx))
                def run(frames):
                    t0 = time.perf_counter()
                    for k in range(frames):
                        r = rs[k % F]; r.set_camera_arrays(*poses[k % 120], fx, fx); r.render_async()
                    for r in rs: r.sync()
                    return frames / (time.perf_counter() - t0)
                run(30)
                out.append("%s/%d %.0f" % ("long" if pol == "1" else "short", F, max(run(240), run(240))))
                st = rs[0].stats()
                for r in rs: r.dispose()
        del os.environ["GSR_LONG_ITEMS"]
        tau = st["tile_entries"] / max(1, st["frames"] if False else 1)
        print("%s n=%7d  D=%d -> layers %.0f | %s" % (gen, n, st["tile_entries"], st["tile_entries"] * 256.0 / (W * H), "  ".join(out)), flush=True)
