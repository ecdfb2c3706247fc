#!/usr/bin/env python3
"""Host-side cost of one N>1 bench step (camera upload + frame enqueue + native exchange calls) with the collective
replaced by a no-op: how many frames/s the Python host can issue per rank."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))
import torch
import gsplat_hip as gh
from gsplat_hip import bands

class NoDist:
    def all_gather_into_tensor(self, flat, slab): pass

cfg = gh.synth.CONFIGS["C1"]          # tiny frames: the device is never the limit
W, H = cfg["width"], cfg["height"]
scene = gh.Scene(); scene.setData(gh.synth.config_rows("C1"))
edges = bands.band_edges(W, 8)
F = 3
rs = [gh.HIPRenderer(W, H, band=edges[3], timing=True, throughput=True) for _ in range(F)]
for r in rs:
    r.render(scene, gh.orbit_camera(0, 120, W, H, cfg["fx"])); r.set_timing_interval(8)
links = [bands.StreamLink(torch, r, "cuda:0") for r in rs]
x = bands.FrameExchange(NoDist(), torch, W, H, 3, 8, torch.device("cuda:0"), edges=edges, dtype=torch.uint8)
poses = [gh.orbit_camera(k, 120, W, H, cfg["fx"]).f32() for k in range(120)]
def step(k, exchange):
    v, p, vp = poses[k % 120]; c = k % F
    rs[c].set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"]); rs[c].render_async()
    if exchange: x.exchange_native(rs[c], links[c])
for mode in (False, True):
    for k in range(100): step(k, mode)
    for r in rs: r.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3000
    for k in range(n): step(k, mode)
    t_issue = time.perf_counter() - t0
    for r in rs: r.sync()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("exchange=%s: host issue %.1f us/frame, wall %.1f us/frame" % (mode, t_issue / n * 1e6, t_all / n * 1e6))
