#!/usr/bin/env python3
"""Does the work-item policy (k_bin_finalize: per bin, from the frame's optical depth per list entry and its tiles per
splat) pick the faster cut?  Scene families beyond the two the thresholds were tuned on -- five single blobs and three
that are not (a tight cluster in a sparse halo, two clusters at different depths, a thin opaque shell: frames that mix
bins that saturate with bins that do not); every bin cut into segments / every bin one item (pinned) vs the automatic
per-bin choice, one frame at a time and three in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import gsplat_hip as gh

W, H, fx = 1920, 1080, 1132.0
poses = [gh.orbit_camera(k, 120, W, H, fx).f32() for k in range(120)]
families = [("big sparse", 150_000, 1.5, 0.03, 0.25), ("big dense", 400_000, 1.2, 0.02, 0.12), ("tiny dense", 2_000_000, 1.0, 0.002, 0.012),
            ("wide thin", 800_000, 2.5, 0.004, 0.05), ("C3-like 1.5M", 1_500_000, 1.5, 0.004, 0.06)]
families += [("cluster+halo", None, gh.synth.cluster_in_halo, 0, 0), ("two clusters", None, gh.synth.two_clusters, 0, 0),
             ("opaque shell", None, gh.synth.opaque_shell, 0, 0)]
only = sys.argv[1:]     # optional: substrings of the family names to run
for name, n, sigma, s_lo, s_hi in families:
    if only and not any(o in name for o in only):
        continue
    rows = gh.synth.synth_rows(n, 11, sigma, s_lo, s_hi) if n else sigma()
    n = rows.size // 32
    scene = gh.Scene(); scene.setData(rows)
    res = {}
    for pol in ("0", "1", None):
        if pol is None: os.environ.pop("GSR_LONG_ITEMS", None)
        else: os.environ["GSR_LONG_ITEMS"] = pol
        for F in (1, 3):
            rs = [gh.HIPRenderer(W, H, throughput=F > 1) for _ in range(F)]
            for r in rs: r.render(scene, gh.orbit_camera(0, 120, W, H, fx))
            def run(frames):
                t0 = time.perf_counter()
                for k in range(frames):
                    r = rs[k % F]; r.set_camera_arrays(*poses[k % 120], fx, fx); r.render_async()
                for r in rs: r.sync()
                return frames / (time.perf_counter() - t0)
            run(30)
            res[(pol, F)] = max(run(180), run(180))
            st = rs[0].stats()
            for r in rs: r.dispose()
    os.environ.pop("GSR_LONG_ITEMS", None)
    lay = st["tile_entries"] * 256.0 / (W * H)
    tps = st["tile_entries"] / max(1, st["visible"])
    line = "%-13s n=%7d layers %5.0f tiles/splat %4.1f |" % (name, n, lay, tps)
    for F in (1, 3):
        s_, l_, a_ = res[("0", F)], res[("1", F)], res[(None, F)]
        best = max(s_, l_)
        line += "  F=%d short %5.0f long %5.0f auto %5.0f (%+.0f%% vs best)" % (F, s_, l_, a_, 100 * (a_ / best - 1))
    print(line, flush=True)
