#!/usr/bin/env python3
"""Per-bin saturation depths of one frame (diagnostic build: scripts/build_exp.sh stamps "-DGSR_BLEND_STAMPS"; whole-bin
work items): entries, entries staged before the item ended, visits per wave, item duration.
  python scripts/bin_depths.py [config] [pose] -> gpurun_out/bin_depths_<config>_<pose>.npy  (rows: bin; cols: 8 words)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import numpy as np
os.environ["GSR_LONG_ITEMS"] = "1"
import gsplat_hip as gh
config = sys.argv[1] if len(sys.argv) > 1 else "C3"
pose = int(sys.argv[2]) if len(sys.argv) > 2 else 35
lib = os.path.join(ROOT, "gsplat.js_amd", "lib_exp", "stamps", "libgsplat_hip.so")
cfg = gh.synth.CONFIGS[config]
W, H = cfg["width"], cfg["height"]
scene = gh.Scene(); scene.setData(gh.synth.config_rows(config))
r = gh.HIPRenderer(W, H, timing=True, lib_path=lib)
r.render(scene, gh.orbit_camera(0, 120, W, H, cfg["fx"]))
r.set_camera(gh.orbit_camera(pose, 120, W, H, cfg["fx"]))
r.reset_stats(); r.render_async(); r.sync()
buf = np.zeros(16384 * 8, dtype=np.uint32)
r._L.gsr_debug_bin_info(ctypes.c_void_p(buf.ctypes.data))
nb = r.work_items()["bins"]
info = buf.reshape(-1, 8)[:nb].copy()
st = r.stats()
out = os.path.join(ROOT, "gpurun_out", "bin_depths_%s_%d.npy" % (config, pose))
np.save(out, info)
c, s = info[:, 0].astype(np.float64), info[:, 1].astype(np.float64)
vis = info[:, 2:6].astype(np.float64)
print("%s pose %d: k_blend %.1f us, %d bins, entries %.3g, staged %.3g (%.0f %%), visits %.3g, max visits/wave %d, longest item %.1f us"
      % (config, pose, st["sum_ms_blend"] / max(1, st["frames"]) * 1e3, nb, c.sum(), s.sum(), 100 * s.sum() / c.sum(), vis.sum(), vis.max(), info[:, 6].max() * 0.01))
order = np.argsort(-info[:, 6].astype(np.int64))[:15]
for b in order:
    print("  bin %5d (%3d,%3d): entries %6d staged %6d visits %s  %.1f us" % (b, b % -(-W // 32), b // -(-W // 32), info[b, 0], info[b, 1], list(info[b, 2:6]), info[b, 6] * 0.01))
r.dispose()
