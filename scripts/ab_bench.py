#!/usr/bin/env python3
"""A/B of library builds in ONE process, interleaved rounds (cdna_hip_programming.md rule 24: per-process and DVFS
variance otherwise looks like a kernel property).

  python scripts/ab_bench.py [--config C3] [--rounds 7] [--frames 40] [--inflight 1] base exp1 exp2 ...

`base` = gsplat.js_amd/lib/libgsplat_hip.so, any other name = gsplat.js_amd/lib_exp/<name>/libgsplat_hip.so
(scripts/build_exp.sh).  Every variant gets its own context(s) on the same scene; a round renders `frames` orbit frames
per variant (one frame in flight: stage times are the uncontended kernel times); the table shows the median over
rounds of each stage's mean time, and frames/s.  --check compares every variant's image and depthIndex with base's.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--inflight", type=int, default=1)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--res-div", type=int, default=1, help="render at 1/N of the configuration's resolution and focal length (same splats, coarser bin grid)")
    ap.add_argument("--no-timing", action="store_true", help="contexts without stage events: frames replay the captured graph (frames/s only)")
    ap.add_argument("--n", type=int, default=0, help="splats (default: the configuration's), same generator parameters")
    ap.add_argument("--sort-only", action="store_true", help="time gsr_sort (key + min/max + radix) instead of full frames")
    ap.add_argument("--exact-contexts", action="store_true", help="contexts without GSR_FLAG_THROUGHPUT even with several frames in flight "
                    "(a variant may still ask for the throughput kind with KIND=throughput among its knobs)")
    args = ap.parse_args()
    import gsplat_hip as gh

    cfg = dict(gh.synth.CONFIGS[args.config])
    cfg["width"] //= args.res_div; cfg["height"] //= args.res_div; cfg["fx"] /= args.res_div
    W, H = cfg["width"], cfg["height"]
    scene = gh.Scene()
    if args.n:
        scene.setData(gh.synth.synth_rows(args.n, cfg["seed"], cfg["sigma"], cfg["s_lo"], cfg["s_hi"]))
    else:
        scene.setData(gh.synth.config_rows(args.config))
    poses = [gh.orbit_camera(k, 120, W, H, cfg["fx"]).f32() for k in range(120)]
    F = max(1, args.inflight)

    def lib_of(name):
        name = name.split("@")[0]
        return None if name == "base" else os.path.join(ROOT, "gsplat.js_amd", "lib_exp", name, "libgsplat_hip.so")

    ctxs = {}
    for name in args.variants:      # "lib@ENV=value,ENV2=value": tuning knobs the library reads when a context is set up
        envs = dict(kv.split("=") for kv in name.split("@")[1].split(",")) if "@" in name else {}
        kind = envs.pop("KIND", None)      # KIND=exact|throughput: the context kind of this variant (not an environment knob)
        os.environ.update(envs)
        tp = (F > 1 and not args.exact_contexts) if kind is None else kind == "throughput"
        rs = []
        for _ in range(F):
            r = gh.HIPRenderer(W, H, timing=not args.no_timing, throughput=tp, lib_path=lib_of(name))
            r.render(scene, gh.orbit_camera(0, 120, W, H, cfg["fx"]))
            rs.append(r)
        for k in envs:
            del os.environ[k]
        ctxs[name] = rs
    if args.check:
        ref = None
        for name, rs in ctxs.items():
            r = rs[0]
            r.set_camera_arrays(*poses[37], cfg["fx"], cfg["fx"])
            r.render_async(); r.sync()
            img, di = r.readPixelsFloat(), r.lastDepthIndex()
            if ref is None:
                ref = (img, di)
            else:
                print("check %-14s depthIndex equal: %s   max |img - base| = %.3g" % (name, np.array_equal(di, ref[1]), float(np.abs(img - ref[0]).max())))
    stages = ("project_key", "sort", "bin", "blend", "combine", "total")
    res = {name: {s: [] for s in stages + ("fps",)} for name in args.variants}
    for rnd in range(args.rounds + 1):
        for name in args.variants:
            rs = ctxs[name]
            for r in rs:
                r.reset_stats()
            t0 = time.perf_counter()
            for k in range(args.frames):
                r = rs[k % F]
                r.set_camera_arrays(*poses[(rnd * 13 + k) % 120], cfg["fx"], cfg["fx"])
                if args.sort_only:
                    r.sort()
                else:
                    r.render_async()
            for r in rs:
                r.sync()
            dt = time.perf_counter() - t0
            if rnd == 0:
                continue   # warm-up round
            st = [r.stats() for r in rs]
            fr = max(1, sum(int(x["frames"]) for x in st))
            for s in stages:
                res[name][s].append(0.0 if args.no_timing else sum(x["sum_ms_" + s] for x in st) / fr * 1e3)
            res[name]["fps"].append(args.frames / dt)
    print("%-26s %8s | %s   (us, median of %d rounds x %d frames, %d in flight, %s)" % (
        "variant", "fps", " ".join("%11s" % s for s in stages), args.rounds, args.frames, F, args.config))
    base = None
    for name in args.variants:
        med = {s: float(np.median(v)) for s, v in res[name].items()}
        if base is None:
            base = med
        print("%-26s %8.1f | %s" % (name, med["fps"], " ".join("%11.1f" % med[s] for s in stages)))
        if med is not base:
            print("%-26s %7.1f%% | %s" % ("  vs base", (med["fps"] / base["fps"] - 1) * 100,
                                           " ".join("%10.1f%%" % ((med[s] / base[s] - 1) * 100 if base[s] else 0) for s in stages)))
    for rs in ctxs.values():
        for r in rs:
            r.dispose()


if __name__ == "__main__":
    main()
