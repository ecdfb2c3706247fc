#!/usr/bin/env python3
"""What a short timed region pays beyond the steady rate: K frames on F contexts between two fences, for several K; the fit
T(K) = a + b K gives the per-frame time b and the fill + drain + host overhead a (the driver times 20 steps).
usage: python scripts/fill_drain.py [--config C3] [--inflight 3] [ENV=v,ENV=v ...]   (each argument = one variant of the environment)"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh

args = sys.argv[1:]
cfgname, F = "C3", 3
while args and args[0].startswith("--"):
    if args[0] == "--config": cfgname = args[1]
    elif args[0] == "--inflight": F = int(args[1])
    args = args[2:]
variants = args or [""]
cfg = gh.synth.CONFIGS[cfgname]
W, H, fx = cfg["width"], cfg["height"], cfg["fx"]
rows = gh.synth.config_rows(cfgname)
scene = gh.Scene(); scene.setData(rows)
poses = [gh.orbit_camera(k, 120, W, H, fx).f32() for k in range(120)]
Ks = (20, 40, 80, 160, 320)
for var in variants:
    env = dict(kv.split("=") for kv in var.split(",") if kv)
    for k, v in env.items(): os.environ[k] = v
    rs = [gh.HIPRenderer(W, H, throughput=F > 1) for _ in range(F)]
    for r in rs: r.render(scene, gh.orbit_camera(0, 120, W, H, fx))
    for r in rs: r.set_timing_interval(0xffffffff)
    def run(K, k0):
        for r in rs: r.sync()
        t0 = time.perf_counter()
        for k in range(K):
            r = rs[k % F]; r.set_camera_arrays(*poses[(k0 + k) % 120], fx, fx); r.render_async()
        for r in rs: r.sync()
        return (time.perf_counter() - t0) * 1e6
    run(30, 0)
    T = {}
    for K in Ks:
        T[K] = statistics.median(run(K, 7 * i) for i in range(9))
    n = len(Ks); sx = sum(Ks); sy = sum(T.values()); sxx = sum(k * k for k in Ks); sxy = sum(k * T[k] for k in Ks)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    print("%-28s %s F=%d | " % (var or "(default)", cfgname, F) + "  ".join("K=%d %.0f us (%.0f fps)" % (k, T[k], k / T[k] * 1e6) for k in Ks) +
          " | fit: %.1f us per frame (%.0f fps) + %.0f us" % (b, 1e6 / b, a), flush=True)
    for r in rs: r.dispose()
    for k in env: os.environ.pop(k, None)
