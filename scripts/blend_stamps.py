#!/usr/bin/env python3
"""Where do the compositor's waves spend their cycles?  Needs the diagnostic build
(scripts/build_exp.sh stamps "-DGSR_BLEND_STAMPS"); prints the shares of wave lifetime per phase of k_blend."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "stamps"
config = sys.argv[2] if len(sys.argv) > 2 else "C3"
lib = os.path.join(ROOT, "gsplat.js_amd", "lib_exp", name, "libgsplat_hip.so")
cfg = gh.synth.CONFIGS[config]
W, H = cfg["width"], cfg["height"]
scene = gh.Scene()
scene.setData(gh.synth.config_rows(config))
r = gh.HIPRenderer(W, H, timing=True, lib_path=lib, early_out_eps=float(os.environ.get("EPS", "0")))
r.render(scene, gh.orbit_camera(0, 120, W, H, cfg["fx"]))
L = r._L
import numpy as np
r.set_camera(gh.orbit_camera(35, 120, W, H, cfg["fx"]))
r.reset_stats()
r.render_async(); r.sync()
buf = np.zeros(4096 * 4 * 16, dtype=np.uint32)
L.gsr_debug_blend_stamps(ctypes.c_void_p(buf.ctypes.data))
st = r.stats()
raw = buf.reshape(-1, 16)
v = raw[:, :8].astype(np.float64)
v = v[v[:, 0] > 0]
life = v[:, 0]
print("%s %s: one frame, k_blend %.1f us (with stamps); %d waves; wave lifetime mean %.0f / max %.0f cycles"
      % (name, config, st["sum_ms_blend"] / max(1, st["frames"]) * 1e3, len(v), life.mean(), life.max()))
if os.environ.get("COUNT_QUADS") == "1":
    q, e, c = v[:, 1].sum(), v[:, 2].sum(), v[:, 3].sum()
    print("quadrant visits %.4g, without a covered pixel %.4g (%.1f %%), covered pixels %.4g = %.1f %% of the lanes of visited quadrants, %.1f %% of the non-empty ones; entry visits %.4g (%.2f quadrants per visit)"
          % (q, e, 100 * e / q, c, 100 * c / (64 * q), 100 * c / (64 * (q - e)), v[:, 7].sum(), q / v[:, 7].sum()))
    r.dispose()
    sys.exit(0)
tot = life.sum()
names = ["barrier A (chunk start)", "staging", "barrier B (staged)", "composite loop"]
for k, nm in enumerate(names):
    print("  %-26s %5.1f %% of wave time" % (nm, 100.0 * v[:, 1 + k].sum() / tot))
print("  %-26s %5.1f %%" % ("other (queue draw, exit)", 100.0 * (tot - v[:, 1:5].sum()) / tot))
print("entry visits %.3g; composite loop %.1f wave-cycles per entry visit (7 waves share a SIMD: x1/7 per SIMD)"
      % (v[:, 7].sum(), v[:, 4].sum() / max(1.0, v[:, 7].sum())))
ok = raw[:, 0] > 0
t0 = raw[ok, 5].astype(np.int64)
t1 = raw[ok, 6].astype(np.int64)
base = t0.min()
s0 = np.sort((t0 - base) & 0xffffffff); e1 = np.sort((t1 - base) & 0xffffffff)
print("wave starts (us after the first wave): p50 %.1f p80 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(0.01 * x for x in (s0[len(s0)//2], s0[int(len(s0)*0.8)], s0[int(len(s0)*0.9)], s0[int(len(s0)*0.99)], s0[-1])))
print("wave ends (us): p1 %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f" % tuple(0.01 * x for x in (e1[int(len(e1)*0.01)], e1[int(len(e1)*0.1)], e1[int(len(e1)*0.25)], e1[len(e1)//2], e1[int(len(e1)*0.75)], e1[int(len(e1)*0.9)], e1[-1])))
dur = ((t1 - t0) & 0xffffffff).astype(np.float64) * 0.01
print("resident wave-time / (7168 wave slots x span) = %.2f" % (dur.sum() / (7168.0 * 0.01 * e1[-1])))
print("idle tail: mean wave lifetime / kernel duration in cycles (at 2.2 GHz) = %.2f" % (life.mean() / (st["sum_ms_blend"] / max(1, st["frames"]) * 1e-3 * 2.2e9)))
w = raw[ok]
order = np.argsort((w[:, 6].astype(np.int64) - base) & 0xffffffff)[::-1]
nbx = -(-W // 32)
print("the 12 waves that ended last: end us | items | last item started us | longest item us, entries, visits/wave, bin (x,y)")
for k in order[:48:4]:
    g0 = (k // 4) * 4
    print("   %7.1f | %3d | %7.1f | %6.1f %5d %s  bin (%d,%d)  composite share of the 4 waves %s" % (
        0.01 * ((int(w[k, 6]) - base) & 0xffffffff), w[k, 8], 0.01 * ((int(w[k, 9]) - base) & 0xffffffff),
        0.01 * w[k, 10], w[k, 11], [int(w[g0 + j, 12]) for j in range(4)], int(w[k, 13]) % nbx, int(w[k, 13]) // nbx,
        ["%.2f" % (w[g0 + j, 4] / max(1.0, float(w[g0 + j, 0]))) for j in range(4)]))
md = np.sort(w[::4, 10].astype(np.float64) * 0.01)
print("longest item per workgroup (us): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (md[int(len(md)*0.1)], md[len(md)//2], md[int(len(md)*0.9)], md[int(len(md)*0.99)], md[-1]))
r.dispose()
