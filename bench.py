#!/usr/bin/env python3
"""Headline benchmark: frames/sec (and sorted-splats/sec) of the gsplat.js hot path
-- depth key + sort + projection + binning + front-to-back composite -- on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N=1: plain process; N>1: torch.distributed.run)

A step is one full frame (renderer.render(scene, camera)) at the next pose of a
120-frame orbit.  Frames are independent, so up to --frames-in-flight (default 3) of
them are in flight on separate contexts/streams: the latency-bound sort/binning
kernels of one frame fill the GPU while another frame composites.  Every one of the
K timed frames is rendered completely inside the timed region.  Workload at every N: BASELINE.json configs[2] = C3, 1M synthetic
gaussians at 1920x1080 (scene bytes resident in HBM before the timed region).
N>1 splits ONE frame across ranks by screen-tile columns and all-gathers the
framebuffer over RCCL: fixed total work, so `scaling` is "strong".
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SETUP_FRAMES = int(__import__("os").environ.get("GSR_BENCH_SETUP_FRAMES", "4"))      # per context, before the warm-up: graph captures and the sort-order decision (see main)
WARM_FRAMES = int(__import__("os").environ.get("GSR_BENCH_WARM_FRAMES", "300"))      # frames rendered (untimed) before the warm-up steps: a device that has been rendering (see main)
ORBIT_FRAMES = 120


def host_cpu():
    """(model name, logical cores of the host, cores this process may use)"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = list(range(os.cpu_count() or 1))
    return model, os.cpu_count() or 1, usable


def cpu_baseline(gh, cfg, data, pos, sample_frames=2, sort_calls=10):
    """The reference's CPU path timed on this host.  Sort: the reference's own wasm/wasm.cpp compiled
    natively (oracle/_ref) when present, else the bit-identical restatement; ONE thread pinned to one core,
    like the reference's single worker (Worker.ts:36-57).  Project + composite have no CPU implementation in
    the reference (they run in WebGL): the oracle's restatement stands in, on `cores` threads."""
    from oracle import oracle as O
    model, host_cores, usable = host_cpu()
    cores = min(len(usable), 16)
    cams = [gh.orbit_camera(k * 7, ORBIT_FRAMES, cfg["width"], cfg["height"], cfg["fx"]) for k in range(max(sort_calls, sample_frames))]
    use_ref = O.ref_available()
    pinned = None
    try:
        os.sched_setaffinity(0, {usable[len(usable) // 2]})   # 1 thread, pinned
        pinned = usable[len(usable) // 2]
    except (AttributeError, OSError):
        pass
    t_sort = []
    try:
        for cam in cams[:2]:    # warm-ups
            (O.ref_sort(cam.f32()[2], pos, calls=1) if use_ref else O.sort(cam.f32()[2], pos))
        for cam in cams[:sort_calls]:
            vp = cam.f32()[2]
            t0 = time.perf_counter()
            if use_ref:
                O.ref_sort(vp, pos, calls=1)  # two reference calls inside (count + sort): halve below
            else:
                O.sort(vp, pos)
            t_sort.append((time.perf_counter() - t0) / (2.0 if use_ref else 1.0))
    finally:
        if pinned is not None:
            os.sched_setaffinity(0, set(usable))
    t_sort = float(np.median(t_sort))
    t_frame = []
    for cam in cams[:sample_frames]:
        v, p, vp = cam.f32()
        t0 = time.perf_counter()
        di, _, _ = O.sort(vp, pos)
        rec, bbox, raw = O.project(data, v, p, cfg["fx"], cfg["fx"], cfg["width"], cfg["height"])
        O.render(di, raw, rec, bbox, cfg["width"], cfg["height"], 1, cores)
        t_frame.append(time.perf_counter() - t0)
    t_frame = float(np.mean(t_frame))
    n = pos.size // 3
    # the same four passes in plain JavaScript under Node (tools/sort_js_baseline.js): what the reference's worker
    # would cost without wasm; reported beside the native build of the reference's C++ (no emcc here to rebuild the wasm)
    js = None
    try:
        import shutil, subprocess, tempfile
        node = shutil.which("node")
        if node:
            with tempfile.NamedTemporaryFile(suffix=".f32") as tf:
                np.asarray(pos, dtype=np.float32).tofile(tf.name)
                vp = cams[0].f32()[2]
                out = subprocess.run([node, os.path.join(ROOT, "tools", "sort_js_baseline.js"), tf.name, repr(float(vp[2])),
                                      repr(float(vp[6])), repr(float(vp[10])), "5"], capture_output=True, text=True, timeout=120)
                js = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        js = None
    # The one native boundary the reference has: wasm `sort` (wasm/wasm.cpp:8-13, called at Worker.ts:36-43).  Its drop-in
    # gsplat_sort_host takes HOST pointers, so a call is pageable H2D of 12 N bytes + the device chain + D2H of 4 N bytes:
    # timed here at three sizes beside the reference's own sort (native build, one pinned core) on the same positions, and
    # beside the device chain alone (gsr_sort on a resident scene, HIP events) -- the difference is transfers and host calls.
    sort_host = {}
    try:
        L = gh.load_library()
        for n_ in (10000, 300000, 1000000):
            if n_ > n:
                continue
            p_ = np.ascontiguousarray(pos[:3 * n_], dtype=np.float32)
            out = np.empty(n_, dtype=np.uint32)
            vps = [np.ascontiguousarray(c.f32()[2], dtype=np.float32) for c in cams[:8]]
            for k in range(3):
                L.gsplat_sort_host(vps[k].ctypes.data, n_, p_.ctypes.data, None, out.ctypes.data, None, None)
            ts = []
            for k in range(12):
                t0 = time.perf_counter()
                L.gsplat_sort_host(vps[k % 8].ctypes.data, n_, p_.ctypes.data, None, out.ctypes.data, None, None)
                ts.append(time.perf_counter() - t0)
            ok = bool(np.array_equal(out, O.sort(vps[11 % 8], p_)[0]))
            tc = []
            try:
                os.sched_setaffinity(0, {usable[len(usable) // 2]})
            except (AttributeError, OSError):
                pass
            try:
                for k in range(6):
                    t0 = time.perf_counter()
                    (O.ref_sort(vps[k], p_, calls=1) if use_ref else O.sort(vps[k], p_))
                    tc.append((time.perf_counter() - t0) / (2.0 if use_ref else 1.0))
            finally:
                try:
                    os.sched_setaffinity(0, set(usable))
                except (AttributeError, OSError):
                    pass
            r = gh.HIPRenderer(64, 64, timing=True)
            r.set_raw_scene(data[:8 * n_], p_)
            for k in range(20):
                if k == 8:
                    r.reset_stats()
                r.set_camera(cams[k % len(cams)])
                r.sort()
            st = r.stats()
            dev_ms = (st["sum_ms_project_key"] + st["sum_ms_sort"]) / max(int(st["frames"]), 1)
            r.dispose()
            m = float(np.median(ts)) * 1e3
            sort_host[str(n_)] = {"sort_host_ms": m, "device_chain_ms": dev_ms, "transfers_and_host_calls_ms": m - dev_ms,
                                  "h2d_bytes": 12 * n_, "d2h_bytes": 4 * n_, "cpu_reference_sort_ms": float(np.median(tc[1:])) * 1e3,
                                  "depth_index_equals_oracle": ok}
    except Exception as e:   # (never fatal for the headline)
        sort_host = {"error": repr(e)}
    return {
        "sort_host": sort_host,
        "sort_host_note": "gsplat_sort_host(viewProj, n, positions, NULL, depthIndex, NULL, NULL) on pageable host arrays, median of 12 calls; "
                          "cpu_reference_sort_ms = the reference's wasm.cpp built natively, one pinned core, same positions",
        "sort_js_ms": js["ms_median"] if js else None,
        "sort_js_note": "wasm.cpp's loops in plain JavaScript (V8, Math.fround per operation), 1 thread; an upper bound for the wasm worker" if js else None,
        "value": 1.0 / t_frame, "unit": "frames/s", "cores": cores, "kind": "port",
        "host_cpu": model, "host_cores_total": host_cores, "sort_pinned_to_cpu": pinned,
        "sample": "%d full frames of the same %d-splat %dx%d workload (sort 1 thread + project 1 thread + composite on %d threads)"
                  % (sample_frames, n, cfg["width"], cfg["height"], cores),
        "sort_splats_per_s": n / t_sort, "sort_ms": t_sort * 1e3, "sort_cores": 1,
        "sort_kind": "reference" if use_ref else "port",
        "sort_sample": "%d sorts of the %d-splat scene after 2 warm-ups, median; 1 core used of %d" % (sort_calls, n, host_cores),
        "frame_ms": t_frame * 1e3,
    }


def other_config(gh, name, device, F, frames=120):
    """The same measurement on another BASELINE.json configuration, outside the headline's timed region: frames/s with F
    frames in flight and one frame at a time, stage times (HIP events, one frame at a time), counts, overflow counters.
    The scene is built on the device from the .splat rows (gsr_set_scene_rows: bit-identical to the host's Scene.setData,
    tests/test_gpu_parity.py), which spares the host's covariance loop for 5 M splats."""
    cfg = gh.synth.CONFIGS[name]
    W, H, N = cfg["width"], cfg["height"], cfg["n"]
    rows = gh.synth.config_rows(name)
    poses = [gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]).f32() for k in range(ORBIT_FRAMES)]
    rs = [gh.HIPRenderer(W, H, device=device, throughput=F > 1, timing=True) for _ in range(F)]
    solo = gh.HIPRenderer(W, H, device=device, timing=True)
    for rr in rs + [solo]:
        rr.set_scene_rows(rows)
        rr.set_timing_interval(8)
        for j in range(SETUP_FRAMES):   # LSD-order first frame, bucket-order decision, graph captures: not a frame's work
            rr.set_camera_arrays(*poses[j], cfg["fx"], cfg["fx"])
            rr.render_async()
        rr.sync()
        rr.reset_stats()
    del rows

    def run(ctxs, count):
        t0 = time.perf_counter()
        for k in range(count):
            rr = ctxs[k % len(ctxs)]
            rr.set_camera_arrays(*poses[k % ORBIT_FRAMES], cfg["fx"], cfg["fx"])
            rr.render_async()
        for rr in ctxs:
            rr.sync()
        return count / (time.perf_counter() - t0)

    run(rs, 2 * F)
    fps = run(rs, frames)
    run([solo], 2)
    solo.reset_stats()
    fps1 = run([solo], max(30, frames // 2))
    s1 = solo.stats()
    f1 = max(int(s1["frames"]), 1)
    sf = max(int(s1["sum_frames"]), 1)
    sts = [rr.stats() for rr in rs + [solo]]
    out = {"workload": "%s: %d synthetic gaussians (seed %d), %dx%d, 120-pose orbit" % (name, N, cfg["seed"], W, H),
           "frames_per_sec": fps, "frames_in_flight": F, "frames": frames,
           "one_frame_in_flight": {"frames_per_sec": fps1, "ms_per_frame": 1e3 / fps1,
                                   "stage_ms": {k: s1["sum_ms_" + k] / f1 for k in ("project_key", "sort", "bin", "blend", "combine", "total")}},
           "counts": {"N": N, "V": s1["sum_visible"] / sf, "D_tiles16": s1["sum_tile_entries"] / sf,
                      "bin_entries32": s1["sum_bin_entries"] / sf, "P": W * H},
           "work_items": solo.work_items(),
           "overflow_frames": sum(int(x["overflow_frames"]) for x in sts), "dropped_frames": sum(int(x["dropped_frames"]) for x in sts)}
    for rr in rs + [solo]:
        rr.dispose()
    return out


def sh_config(gh, device, F, frames=120):
    """C3 with degree-3 spherical harmonics on EVERY splat (the fork's raison d'etre: vertex.glsl.ts:57-104,180-204,
    Scene.ts:108-124): three half textures of 8 words per splat (96 bytes per splat more for the projection kernel to read,
    inverse(view) and the degree-3 polynomial per visible splat, a float4 colour written per visible splat and read by the
    compositor's staging instead of the record's rgb8).  Seeded coefficients: the half words are drawn directly (sign,
    exponent 2^-6..2^-3, mantissa), which is what Scene.setData's packHalf2x16 would have produced from such floats."""
    name = "C3"
    cfg = gh.synth.CONFIGS[name]
    W, H, N = cfg["width"], cfg["height"], cfg["n"]
    rows = gh.synth.config_rows(name)
    rng = np.random.default_rng(33)
    tex = []
    for _ in range(3):
        w = rng.integers(0, 1 << 32, size=8 * N, dtype=np.uint64).astype(np.uint32)
        ex = rng.integers(9, 13, size=8 * N, dtype=np.uint32)
        ex2 = rng.integers(9, 13, size=8 * N, dtype=np.uint32)
        w = (w & np.uint32(0x83FF83FF)) | (ex << np.uint32(10)) | (ex2 << np.uint32(26))
        tex.append(w)
    bands_idx = np.array([-1, -1, -1], dtype=np.int32)     # every splat: degree 3
    poses = [gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]).f32() for k in range(ORBIT_FRAMES)]
    rs = [gh.HIPRenderer(W, H, device=device, throughput=F > 1, timing=True) for _ in range(F)]
    solo = gh.HIPRenderer(W, H, device=device, timing=True)
    for rr in rs + [solo]:
        rr.set_scene_rows(rows)
        rr.set_sh(tex, bands_idx)
        rr.set_timing_interval(8)
        for j in range(SETUP_FRAMES):
            rr.set_camera_arrays(*poses[j], cfg["fx"], cfg["fx"])
            rr.render_async()
        rr.sync()
        rr.reset_stats()

    def run(ctxs, count):
        t0 = time.perf_counter()
        for k in range(count):
            rr = ctxs[k % len(ctxs)]
            rr.set_camera_arrays(*poses[k % ORBIT_FRAMES], cfg["fx"], cfg["fx"])
            rr.render_async()
        for rr in ctxs:
            rr.sync()
        return count / (time.perf_counter() - t0)

    run(rs, 2 * F)
    fps = run(rs, frames)
    run([solo], 2)
    solo.reset_stats()
    fps1 = run([solo], max(30, frames // 2))
    s1 = solo.stats()
    f1 = max(int(s1["frames"]), 1)
    sf = max(int(s1["sum_frames"]), 1)
    V = s1["sum_visible"] / sf
    pk = s1["sum_ms_project_key"] / f1
    out = {"workload": "C3 with degree-3 SH on every splat: %d synthetic gaussians (seed %d) + 3 x 8 half-packed words per splat (seed 33), %dx%d, 120-pose orbit"
                       % (N, cfg["seed"], W, H),
           "frames_per_sec": fps, "frames_in_flight": F,
           "one_frame_in_flight": {"frames_per_sec": fps1, "ms_per_frame": 1e3 / fps1,
                                   "stage_ms": {k: s1["sum_ms_" + k] / f1 for k in ("project_key", "sort", "bin", "blend", "total")}},
           "project_key": {"ms": pk, "algorithmic_bytes": 32.0 * N + 48.0 * V + (96.0 + 16.0) * V,
                           "GBps": (32.0 * N + 48.0 * V + 112.0 * V) / (pk * 1e-3) / 1e9 if pk else 0,
                           "note": "the plain kernel's 32 N + 48 V plus 96 B of SH halves read and 16 B of colour written per visible splat"},
           "counts": {"N": N, "V": V},
           "overflow_frames": sum(int(x.stats()["overflow_frames"]) for x in rs + [solo])}
    for rr in rs + [solo]:
        rr.dispose()
    return out


def scene_build_config(gh, device):
    """Scene.setData and Scene.rotate (Scene.ts:126-257) as device kernels (gsr_set_scene_rows, gsr_scene_rotate: k_scene.hip,
    every word bit-identical to the host loops) at C4's 5 M splats, beside the same loops in JavaScript (the package's Scene,
    Node, one thread) on a bounded sample of 1 M splats."""
    import shutil
    import subprocess
    rows = gh.synth.config_rows("C4")
    n = rows.size // 32
    r = gh.HIPRenderer(64, 64, device=device)
    r.set_scene_rows(rows[:32 * 1000])          # (context and kernels warm)
    t0 = time.perf_counter()
    r.set_scene_rows(rows)
    t_set = time.perf_counter() - t0
    q = np.array([0.3, -0.2, 0.1, 0.9]); q = q / np.sqrt((q * q).sum())
    ts = {}
    for what, fn in (("rotate", lambda: r.scene_rotate(q)), ("translate", lambda: r.scene_translate([0.25, -0.5, 1.0])),
                     ("scale", lambda: r.scene_scale([1.5, 0.75, 1.25]))):
        fn(); r.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        r.sync()
        ts[what] = (time.perf_counter() - t0) / 5 * 1e3
    r.dispose()
    out = {"device": {"n": n, "set_scene_rows_ms": t_set * 1e3, "set_scene_rows_note": "includes the pageable upload of 32 B per splat",
                      "scene_rotate_ms": ts["rotate"], "scene_translate_ms": ts["translate"], "scene_scale_ms": ts["scale"],
                      "rotate_bytes": (32 + 36) * n * 1.0}}
    node = shutil.which("node")
    if node:
        try:
            o = subprocess.run([node, os.path.join(ROOT, "tools", "scene_js_baseline.js"), "1000000"], capture_output=True, text=True, timeout=120)
            out["js_host"] = json.loads(o.stdout.strip().splitlines()[-1])
        except Exception as e:
            out["js_host"] = {"error": repr(e)}
    return out


def split_config(gh, torch, dist, name, rank, world, local_rank, F, steps, warmup, custom_collective):
    """N>1: configuration `name` split over the ranks by tile columns with the library's exchange (RCCL; a host-staged gloo
    all-gather in the one-GPU rehearsal): BASELINE C5 = C4's scene on N GPUs.  Every rank calls this; the frame rate is the
    whole job's (barrier, MAX over ranks), like the headline's."""
    from gsplat_hip import bands
    cfg = gh.synth.CONFIGS[name]
    W, H, N = cfg["width"], cfg["height"], cfg["n"]
    scene = gh.Scene()
    scene.setData(gh.synth.config_rows(name))
    dev = torch.device("cuda", local_rank)
    cal = gh.HIPRenderer(W, H, device=local_rank)
    cost = np.zeros(-(-W // 32))
    for k in (0, 60):
        cal.render(scene, gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]))
        cost += cal.bin_totals().sum(axis=0) + 0.25 * 32 * H
    cal.dispose()
    got = [bands.balanced_edges(W, world, cost)]
    dist.broadcast_object_list(got, src=0)        # every rank uses rank 0's edges
    edges = got[0]
    rs = [gh.HIPRenderer(W, H, device=local_rank, timing=False, throughput=F > 1) for _ in range(F)]
    if custom_collective:
        def allgather(send, recv, nbytes, stream):
            st = torch.cuda.ExternalStream(stream, device=dev)
            st.synchronize()
            mine = torch.as_tensor(bands.DevicePointer(send, (nbytes,), "|u1"), device=dev).cpu()
            every = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(every, mine)
            with torch.cuda.stream(st):
                torch.as_tensor(bands.DevicePointer(recv, (world * nbytes,), "|u1"), device=dev).copy_(every)
            st.synchronize()
        rs[0].join_group_custom(rank, world, edges, allgather)
    else:
        ids = [gh.new_group_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        rs[0].join_group(ids[0], rank, world, edges)
    for rr in rs[1:]:
        rr.share_group(rs[0])
    poses = [gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]).f32() for k in range(ORBIT_FRAMES)]
    for rr in rs:
        rr.render(scene, gh.orbit_camera(0, ORBIT_FRAMES, W, H, cfg["fx"]))

    def run(k0, count):
        for k in range(k0, k0 + count):
            rr = rs[k % F]
            rr.set_camera_arrays(*poses[k % ORBIT_FRAMES], cfg["fx"], cfg["fx"])
            rr.render_async()
            rr.allgather_frame_async()
        for rr in rs:
            rr.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    run(0, warmup)
    t0 = time.perf_counter()
    run(warmup, steps)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    sts = [rr.stats() for rr in rs]
    out = {"workload": "%s scene split over %d GPUs by tile columns (BASELINE C5): %d synthetic gaussians (seed %d), %dx%d, 120-pose orbit"
                       % (name, world, N, cfg["seed"], W, H),
           "frames_per_sec": steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "n_gpus": world, "steps": steps, "warmup": warmup,
           "frames_in_flight": F, "scaling": "strong", "exchange": "host-staged gloo all-gather (rehearsal)" if custom_collective else
           "RGBA8 slabs, ncclAllGather issued by the library", "band_edges": [list(e) for e in edges],
           "overflow_frames_this_rank": sum(int(x["overflow_frames"]) for x in sts)}
    for rr in reversed(rs):
        rr.dispose()
    return out



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--early-out-eps", type=float, default=0.0,
                    help="0 = no approximate termination (default): every fragment that can change a bit of the image is composited")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the secondary measurements on C2 (300 k, 1080p) and C4 (5 M, 3840x2160) that follow the C3 legs")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the untimed secondary legs (one frame in flight, latency, readback, sort alone) and the CPU baseline: "
                         "for profiler runs, so that every traced launch belongs to the timed configuration")
    ap.add_argument("--exchange", choices=("rgba8", "f32"), default="rgba8",
                    help="N>1: framebuffer slab format of the per-frame all-gather (rgba8: what a display consumes)")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frames rendered concurrently on separate contexts/streams (frame k uses context k mod F)")
    ap.add_argument("--throughput-contexts", action="store_true",
                    help="contexts of the throughput kind (GSR_FLAG_THROUGHPUT: what the default three-in-flight run uses) even with "
                         "--frames-in-flight 1: the timed region's own compositor kernel (k_blend) one launch at a time, for profiler runs")
    ap.add_argument("--equal-bands", action="store_true", help="N>1: equal-width bands instead of cost-balanced ones")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="N>1: nccl = RCCL over xGMI (the product path); gloo = rehearsal on a box without peers: ranks may share "
                         "a GPU and the slab all-gather is staged through host memory")
    ap.add_argument("--timing-interval", type=int, default=8,
                    help="stage events (HIP) on every k-th frame of the timed region; 1 = every frame")
    ap.add_argument("--c5-rehearsal", action="store_true",
                    help="N>1 over gloo: also run the C5 leg (5 M splats at 3840x2160 split over the ranks) with the library's exchange "
                         "driven through a host-staged gloo all-gather (gsr_comm_init_custom) -- a rehearsal of its control flow on one GPU")
    ap.add_argument("--emulate-rank", default=None, metavar="Q/G",
                    help="single GPU only: render just the band rank Q of G would own (no exchange) -> per-rank device time of a G-GPU run")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not under a launcher: start one rank per GPU ourselves.  This process has not touched the GPU (torch is not
        # even imported yet), so the ranks are plain children and this parent only relays their exit code; rank 0
        # prints the JSON line.  (The driver's own form, `python -m torch.distributed.run ... bench.py --gpus N`,
        # arrives with WORLD_SIZE set and takes the branch below.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:   # every value, 1 included: a silent one-GPU run under an N-rank launcher measures nothing
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch
    import gsplat_hip as gh

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()      # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    cfg = gh.synth.CONFIGS[args.config]
    W, H, N = cfg["width"], cfg["height"], cfg["n"]
    rows = gh.synth.config_rows(args.config)
    scene = gh.Scene()
    scene.setData(rows)

    # tile-column bands: whole 32-px bins per rank.  Equal-width bands are badly unbalanced for centre-heavy scenes
    # (a frame is as slow as its slowest band), so the edges come from a calibration frame's per-column list entries.
    from gsplat_hip import bands
    emu = None
    if args.emulate_rank:
        if world > 1:
            raise SystemExit("--emulate-rank is a single-GPU diagnostic")
        emu = tuple(int(x) for x in args.emulate_rank.split("/"))
    eworld = emu[1] if emu else world
    edges = bands.band_edges(W, eworld)
    if eworld > 1 and not args.equal_bands:
        cal = gh.HIPRenderer(W, H, device=local_rank)
        cost = np.zeros(-(-W // 32))
        for k in (0, 30, 60, 90):
            cal.render(scene, gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]))
            cost += cal.bin_totals().sum(axis=0) + 0.25 * 32 * H    # list entries + a per-pixel output term
        cal.dispose()
        got = [bands.balanced_edges(W, eworld, cost)]
        if world > 1:
            dist.broadcast_object_list(got, src=0)    # every rank uses rank 0's edges
        edges = got[0]
    x0, x1 = edges[emu[0] if emu else rank]
    if world > 1 and x1 <= x0:
        raise SystemExit("rank %d has an empty band (more ranks than 32-px bin columns)" % rank)
    band = (x0, x1) if eworld > 1 else None
    F = max(1, args.frames_in_flight)
    # N>1 over RCCL: the exchange lives in the library (gsr_comm_init / gsr_allgather_frame_async), torch.distributed is
    # only the launcher's rendezvous: it carries the communicator ids once and does the barrier / max-over-ranks of the
    # timing.  N>1 over gloo is the rehearsal on a box without peers: ranks may share a GPU (RCCL refuses that), so the
    # slabs go through the harness (bands.FrameExchange, host staged).
    in_library = world > 1 and args.backend == "nccl"
    if in_library and args.exchange != "rgba8":
        raise SystemExit("bench.py: the in-library RCCL exchange ships RGBA8 slabs; --exchange f32 exists only for the gloo rehearsal")
    rs = []
    for _ in range(F):   # every context owns its buffers and its stream; frames are independent of each other
        rr = gh.HIPRenderer(W, H, device=local_rank, early_out_eps=args.early_out_eps, band=None if in_library else band,
                            timing=True, throughput=F > 1 or args.throughput_contexts)
        rs.append(rr)
    if in_library:
        # ONE communicator and ONE exchange stream per rank, shared by its F contexts: the collectives of a rank's frames in
        # flight are issued in frame order on every rank (several communicators per device with collectives in flight on
        # separate streams can deadlock when ranks schedule them in different orders)
        ids = [gh.new_group_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        rs[0].join_group(ids[0], rank, world, edges)
        for rr in rs[1:]:
            rr.share_group(rs[0])
    for rr in rs:
        rr.render(scene, gh.orbit_camera(0, ORBIT_FRAMES, W, H, cfg["fx"]))  # uploads the scene, first frame
        rr.set_timing_interval(max(1, args.timing_interval))
    # One-time work of a context that is not a frame's work: the first frame of a scene sorts in LSD order and reports its
    # largest bucket, the second switches to the bucket order, and each of the two kernel chains is captured into a HIP graph
    # on first use (~0.3 ms of host time each).  With the driver's short runs (--warmup 5 over three contexts) those
    # captures fell into the timed region.  SETUP_FRAMES frames per context settle them before the warm-up steps; the
    # warm-up and the K timed steps that follow are unchanged.
    if world == 1:
        for j in range(SETUP_FRAMES):
            for c, rr in enumerate(rs):
                rr.set_camera(gh.orbit_camera(1 + j * F + c, ORBIT_FRAMES, W, H, cfg["fx"]))
                rr.render_async()
            for rr in rs:
                rr.sync()
    r = rs[0]

    fbs = links = xchg = None
    if world > 1 and not in_library:
        dev = "cuda:%d" % local_rank
        if args.exchange == "rgba8":
            fbs = [bands.framebuffer8_tensor(torch, rr, dev) for rr in rs]
            xchg = bands.FrameExchange(dist, torch, W, H, rank, world, fbs[0].device, edges=edges, dtype=torch.uint8,
                                       host_staged=args.backend == "gloo")
        else:
            fbs = [bands.framebuffer_tensor(torch, rr, dev) for rr in rs]
            xchg = bands.FrameExchange(dist, torch, W, H, rank, world, fbs[0].device, edges=edges, host_staged=args.backend == "gloo")
        links = [bands.StreamLink(torch, rr, dev) for rr in rs]

    # Camera.update for the 120 poses is host JS/Python f64 work outside the device path: precomputed
    poses = [gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]).f32() for k in range(ORBIT_FRAMES)]

    def step(k):
        v, p, vp = poses[k % ORBIT_FRAMES]
        c = k % F
        rr = rs[c]
        rr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
        rr.render_async()
        if in_library:
            rr.allgather_frame_async()   # pack -> ncclAllGather -> de-slab on the context's exchange stream, device-ordered
        elif world > 1:
            # device-side ordering only (no host round trip): the collective and the de-slab on torch's stream overlap the
            # following frames' projection, sort, binning and compositing on the renderers' streams
            if args.exchange == "rgba8":
                xchg.exchange_native(rr, links[c])
            else:
                links[c].torch_waits_for_renderer()
                xchg.exchange(fbs[c])
                # this context's next frame may start as soon as this band has left the framebuffer
                links[c].renderer_waits_for_event(xchg.copied)

    def fence():
        for rr in rs:
            try:
                rr.sync()
            except gh.GsplatError as e:   # frames lost to a list overflow: counted below, the run is then invalid
                if "not composited" not in str(e):
                    raise
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The timed region renders K complete frames and nothing else: no stage events in it (every event is a packet for the
    # command processor, and a frame that carries them is issued as individual launches instead of one graph replay).  The
    # stage times of the same frames under the same overlap are sampled in a pass of their own right behind it (untimed).
    overflow_before = (sum(int(rr.stats()["overflow_frames"]) for rr in rs), sum(int(rr.stats()["dropped_frames"]) for rr in rs))
    for rr in rs:
        rr.set_timing_interval(0xffffffff)
    def timed(k0):   # W untimed warm-up steps, fence, exactly K timed steps, fence
        for k in range(args.warmup):
            step(k0 + k)
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k0 + args.warmup + k)
        fence()
        return time.perf_counter() - t0
    # The device itself warms up too: right after the process's set-up the first few thousand microseconds of frames run ~7 %
    # slower than the same frames 50 ms later (clocks, page tables; scripts/fill_drain.py and DESIGN 8.0 have the numbers), and
    # W = 5 steps are 1 ms.  `value` is the rate of a device that has been rendering (WARM_FRAMES frames, untimed, then the W
    # warm-up steps and the K timed steps as the contract says); the same W + K right after the set-up is kept beside it
    # (`timed_region.device_just_started`), so that both are on record.
    cold_elapsed = timed(0) if WARM_FRAMES else None
    for k in range(WARM_FRAMES):
        step(k)
    elapsed = timed(0)
    for rr in rs:
        rr.set_timing_interval(max(1, args.timing_interval))
        rr.reset_stats()
    for k in range(max(24, 3 * max(1, args.timing_interval))):
        step(args.warmup + args.steps + k)
    fence()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N>1: the 4K scene split over the same ranks (BASELINE C5) in the same run; every rank takes part
    c5 = None
    if world > 1 and args.config == "C3" and not args.no_other_configs and not args.timed_only and (in_library or args.c5_rehearsal):
        c5 = split_config(gh, torch, dist, "C4", rank, world, local_rank, F, max(30, args.steps // 4), 9, custom_collective=not in_library)

    sts = [rr.stats() for rr in rs]
    overflow_frames = sum(int(x["overflow_frames"]) for x in sts) - overflow_before[0]
    dropped_frames = sum(int(x["dropped_frames"]) for x in sts) - overflow_before[1]
    if dropped_frames:
        # a frame whose lists did not fit is not composited; a timed region containing one has not rendered K frames
        raise SystemExit("bench.py: %d frame(s) of the timed region were dropped by a bin-list overflow: result invalid" % dropped_frames)
    st = {k: sum(x[k] for x in sts) for k in sts[0] if k.startswith("sum_") or k == "frames"}
    frames = max(int(st["frames"]), 1)
    ms = {k: st["sum_ms_" + k] / frames for k in ("project_key", "sort", "bin", "blend", "combine", "total")}
    sf = max(int(st["sum_frames"]), 1)
    V, D, E = st["sum_visible"] / sf, st["sum_tile_entries"] / sf, st["sum_bin_entries"] / sf
    band_px = ((x1 - x0) if eworld > 1 else W) * H
    band_edges_used = edges if eworld > 1 else None

    # secondary leg outside the timed region: ONE frame in flight on a context tuned for that (no GSR_FLAG_THROUGHPUT)
    # -> per-frame latency and uncontended stage times
    solo = None
    r_solo_sub = 1
    if F > 1 and world == 1 and not args.timed_only:
        sr = gh.HIPRenderer(W, H, device=local_rank, early_out_eps=args.early_out_eps, timing=True)
        for k in range(4):
            sr.render(scene, gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]))
        r_solo_sub = sr.work_items()["waves_per_tile"]
        sr.reset_stats()
        t1 = time.perf_counter()
        for k in range(60):
            v, p, vp = poses[k % ORBIT_FRAMES]
            sr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
            sr.render_async()
        sr.sync()
        dt = time.perf_counter() - t1
        s1 = sr.stats()
        f1 = max(int(s1["frames"]), 1)
        solo = {"frames_per_sec": 60 / dt, "ms_per_frame": dt / 60 * 1e3,
                "stage_ms": {k: s1["sum_ms_" + k] / f1 for k in ("project_key", "sort", "bin", "blend", "combine", "total")}}
        # host-visible latency of a frame issued on an idle GPU and waited for (render_async + sync), one orbit
        lat = []
        for k in range(ORBIT_FRAMES):
            v, p, vp = poses[k]
            t2 = time.perf_counter()
            sr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
            sr.render_async()
            sr.sync()
            lat.append((time.perf_counter() - t2) * 1e3)
        lat.sort()
        solo["frame_latency_ms"] = {"p50": lat[len(lat) // 2], "p99": lat[min(len(lat) - 1, int(len(lat) * 0.99))],
                                    "mean": sum(lat) / len(lat), "frames": len(lat)}
        # frames/s when the host reads every frame back as RGBA8 (conversion kernel + 8.3 MB D2H into pageable memory)
        t3 = time.perf_counter()
        for k in range(30):
            v, p, vp = poses[k]
            sr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
            sr.render_async()
            sr.readPixels()
        solo["frames_per_sec_with_rgba8_readback"] = 30 / (time.perf_counter() - t3)
        # For the record, the same frames with the compositor's saturation skip switched off (GSR_SATURATE=0: every list
        # entry is visited; the image is the same bit for bit, tests/test_gpu_parity.py): F contexts in flight and one.
        if args.early_out_eps == 0.0:
            os.environ["GSR_SATURATE"] = "0"
            try:
                noskip = {}
                for nctx, key in ((F, "frames_per_sec"), (1, "frames_per_sec_one_frame_in_flight")):
                    xs = [gh.HIPRenderer(W, H, device=local_rank, throughput=nctx > 1) for _ in range(nctx)]
                    for x in xs:
                        for j in range(3):
                            x.render(scene, gh.orbit_camera(j, ORBIT_FRAMES, W, H, cfg["fx"]))
                    t4 = time.perf_counter()
                    nfr = 90
                    for k in range(nfr):
                        v, p, vp = poses[k % ORBIT_FRAMES]
                        x = xs[k % nctx]
                        x.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
                        x.render_async()
                    for x in xs:
                        x.sync()
                    noskip[key] = nfr / (time.perf_counter() - t4)
                    for x in xs:
                        x.dispose()
                # and the claim itself, checked in this run: same cut of the lists (long work items pinned), skip off vs on
                os.environ["GSR_LONG_ITEMS"] = "1"
                a = gh.HIPRenderer(W, H, device=local_rank)
                del os.environ["GSR_SATURATE"]
                b = gh.HIPRenderer(W, H, device=local_rank)
                same = True
                for k in (17, 71):
                    cam_k = gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"])
                    a.render(scene, cam_k)
                    b.render(scene, cam_k)
                    same = same and bool(np.array_equal(a.readPixelsFloat(), b.readPixelsFloat()))
                a.dispose(); b.dispose()
                noskip["image_bit_identical_to_skip_on"] = same
                noskip["note"] = "GSR_SATURATE=0: quadrants whose pixels can no longer change are still visited; identical image"
                solo["without_saturation_skip"] = noskip
            finally:
                os.environ.pop("GSR_SATURATE", None)
                os.environ.pop("GSR_LONG_ITEMS", None)
        # the sort path alone, as the reference's worker runs it (wasm.cpp sort(): key + min/max + quantise + order):
        # gsr_sort = key kernel without projection + the two radix passes
        sr.set_timing_interval(1)          # every call carries its stage events
        for k in range(16 + 96):           # (16 untimed calls first: the clocks of an idle device ramp up over the first few)
            if k == 16:
                sr.reset_stats()
            v, p, vp = poses[k % ORBIT_FRAMES]
            sr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
            sr.sort()
        s2 = sr.stats()
        f2 = max(int(s2["frames"]), 1)
        t_key, t_sort = s2["sum_ms_project_key"] / f2, s2["sum_ms_sort"] / f2
        solo["sort_only"] = {"ms_key_minmax": t_key, "ms_quantise_radix": t_sort,
                             "sorted_splats_per_sec": N / ((t_key + t_sort) * 1e-3),
                             "GBps": 52.0 * N / ((t_key + t_sort) * 1e-3) / 1e9,
                             "frac_of_hbm_peak": 52.0 * N / ((t_key + t_sort) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "HIP events around the whole device chain of gsr_sort: k_depth_key (camera by value, no k_begin_frame "
                                     "in front), k_quantise_hist, column scan, the radix kernels"}
        sr.dispose()
        # The timed region's own compositor kernel, one launch at a time: a context of the kind the timed region uses
        # (GSR_FLAG_THROUGHPUT: k_blend, one wave per tile, its cut of the lists), frames issued one after the other, HIP events
        # on the library's stream.  `roofline` below is this kernel's (the headline and its roofline must be the
        # same kernel); the one-frame context's k_blend2 keeps its own figures in `one_frame_in_flight`.
        tr = gh.HIPRenderer(W, H, device=local_rank, early_out_eps=args.early_out_eps, timing=True, throughput=True)
        for k in range(SETUP_FRAMES):
            tr.render(scene, gh.orbit_camera(k, ORBIT_FRAMES, W, H, cfg["fx"]))
        tr.set_timing_interval(1)
        tr.reset_stats()
        for k in range(60):
            v, p, vp = poses[k % ORBIT_FRAMES]
            tr.set_camera_arrays(v, p, vp, cfg["fx"], cfg["fx"])
            tr.render_async()
            tr.sync()
        s3 = tr.stats()
        f3 = max(int(s3["frames"]), 1)
        solo["timed_region_kernels_one_launch_at_a_time"] = {
            "stage_ms": {k: s3["sum_ms_" + k] / f3 for k in ("project_key", "sort", "bin", "blend", "total")},
            "work_items": tr.work_items(),
            "note": "a GSR_FLAG_THROUGHPUT context (the kind the timed region renders on) rendering one frame at a time"}
        tr.dispose()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        fps = args.steps / elapsed
        # algorithmic bytes per frame, SURVEY.md 8(d): B_sort = 52N, B_proj = 16N + 48V,
        # B_bin = 8D, B_blend = 32D + 16P (per launch of the compositor, this rank's band)
        b_blend = 32.0 * D + 16.0 * band_px
        b_sort = 52.0 * N
        b_proj = 16.0 * N + 48.0 * V
        b_bin = 8.0 * D
        # HBM traffic and VALU instruction counts of k_blend come from rocprofv3 --pmc passes (scripts/gpu.sh pmc ->
        # profiles/blend_traffic.json).  They describe ONE build of the kernels: the file carries the build id it was
        # measured on, and a library with another id gets null rather than a stale figure.
        traffic = valu = traffic_solo = valu_solo = None
        tpath = os.path.join(ROOT, "profiles", "blend_traffic.json")
        traffic_note = "no PMC measurement committed for this workload/configuration"
        if os.path.exists(tpath) and args.config == "C3" and eworld == 1 and args.early_out_eps == 0.0:
            try:
                tj = json.load(open(tpath))
                if tj.get("build_id") == gh.build_id():
                    sel = tj["frames_in_flight" if (F > 1 or args.throughput_contexts) else "one_frame"]
                    traffic, valu = sel["hbm_bytes_per_launch"], sel["valu_wave_instructions_per_launch"]
                    traffic_solo = tj["one_frame"]["hbm_bytes_per_launch"]
                    valu_solo = tj["one_frame"]["valu_wave_instructions_per_launch"]
                    traffic_note = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes on this build (profiles/blend_traffic.json)"
                else:
                    traffic_note = "profiles/blend_traffic.json was measured on build %s, this library is %s: not reported" % (
                        tj.get("build_id"), gh.build_id())
            except Exception:
                traffic = valu = traffic_solo = valu_solo = None
        # The kernel's launch duration: HIP events on the library's stream.  With several frames in flight the kernels of
        # different frames overlap, so an event pair around k_blend also spans time in which other frames' kernels held
        # the CUs: that figure can exceed ms_per_step and is kept only as `timed_region`.  The roofline fraction uses
        # the duration of the same kernel on the same frames with ONE frame in flight (untimed secondary leg), which is
        # what the committed rocprofv3 --kernel-trace summary of `bench.py --frames-in-flight 1 --timed-only` shows too.
        ach = b_blend / (ms["blend"] * 1e-3) / 1e9 if ms["blend"] > 0 else 0.0
        # the timed region's compositor kernel (k_blend on throughput contexts) in isolation; without the secondary legs
        # (--timed-only, N>1) the timed region's own event pairs
        tp = solo["timed_region_kernels_one_launch_at_a_time"] if solo else None
        blend_ms = tp["stage_ms"]["blend"] if tp else ms["blend"]
        ach_solo = b_blend / (blend_ms * 1e-3) / 1e9 if blend_ms > 0 else 0.0
        sm = solo["stage_ms"] if solo else ms   # per-stage figures: uncontended times when several frames were in flight
        timed_sub = rs[0].work_items()["waves_per_tile"]
        kernel_name = "k_blend2" if timed_sub == 2 else "k_blend"
        if solo:   # the one-frame context's compositor (k_blend2) against the same byte model
            b2 = solo["stage_ms"]["blend"]
            solo["roofline"] = {"kernel": "k_blend2" if r_solo_sub == 2 else "k_blend", "avg_launch_ms": b2,
                                "achieved": b_blend / (b2 * 1e-3) / 1e9 if b2 > 0 else 0.0,
                                "frac": b_blend / (b2 * 1e-3) / 1e9 / HBM_PEAK_GBS if b2 > 0 else 0.0,
                                "traffic": traffic_solo,
                                "valu_frac": (valu_solo / (b2 * 1e-3) / 0.651e12) if (valu_solo and b2 > 0) else None}
        out = {
            "metric": "frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d synthetic gaussians (seed %d), %dx%d, 120-pose orbit, full render(scene,camera) "
                                   "= depth key + 17-bit sort + projection + binning + composite"
                                   % (args.config, N, cfg["seed"], W, H),
                       "early_out_eps": args.early_out_eps, "frames_in_flight": F, "stage_events_every": max(1, args.timing_interval), "emulated_rank": args.emulate_rank, "backend": (args.backend if world > 1 else None), "world_size": world, "parallelism": "tile-column bands x%d%s" % (world, "" if world == 1 else (", %s all-gather %s, %s edges" % (args.exchange, "inside the library (RCCL)" if in_library else "in the harness (gloo rehearsal, host staged)", "equal" if args.equal_bands else "cost-balanced"))),
                       "output": "RGBA f32 premultiplied, left in HBM",
                       "saturation_skip": "on (default): a quadrant is no longer visited once no later splat can change a bit of its "
                                          "pixels; image bit-identical to GSR_SATURATE=0, whose rate is in one_frame_in_flight.without_saturation_skip"},
            "sorted_splats_per_sec": N / ((sm["project_key"] + sm["sort"]) * 1e-3) if (sm["project_key"] + sm["sort"]) > 0 else None,
            "stage_ms": ms,
            "counts": {"N": N, "V": V, "D_tiles16": D, "bin_entries32": E, "P": band_px},
            "roofline": {"bound": "valu", "kernel": kernel_name, "achieved": ach_solo, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_solo / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": b_blend, "avg_launch_ms": blend_ms,
                         "frames_in_flight_of_this_figure": 1 if solo else F,
                         "timed_region": None if not solo else {
                             "frames_in_flight": F, "avg_launch_ms": ms["blend"], "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                             "traffic": traffic,
                             "note": "event pairs around k_blend while other frames' kernels share the GPU: overlap-inflated"},
                         "note": "achieved / peak / frac / traffic are the HBM figures the contract asks for (algorithmic bytes over the "
                                 "launch duration); the kernel itself is bound by VALU issue and by a wave's serial walk over its "
                                 "entries, not by HBM (SURVEY 8(d) honest note): see `valu` and DESIGN.md section 8"},
            # secondary ceiling: VALU issue.  peak = what tools/valu_cost.hip sustains on this chip for the compositor's
            # own instruction mix (11 VALU of a covered quadrant incl. v_exp_f32 and v_pk_fma_f32, operands in VGPRs,
            # 7 waves/SIMD like k_blend, wall clock): 0.651e12 wave-instr/s (profiles/r02_valu_cost4.txt)
            "valu": None if not valu or not blend_ms else (lambda t_ms, v: {
                "wave_instr_per_launch": v, "launch_ms": t_ms, "achieved_wave_instr_per_s": v / (t_ms * 1e-3),
                "peak_wave_instr_per_s": 0.651e12, "frac": v / (t_ms * 1e-3) / 0.651e12})(blend_ms, valu),
            "overflow_frames": overflow_frames, "dropped_frames": dropped_frames,
            # the same W + K steps timed right after the process's set-up, before the WARM_FRAMES untimed frames (see main)
            "device_just_started": None if cold_elapsed is None else {
                "frames_per_sec": args.steps / cold_elapsed, "ms_per_step": cold_elapsed / args.steps * 1e3,
                "frames_rendered_before": F * (1 + SETUP_FRAMES), "frames_rendered_before_value": F * (1 + SETUP_FRAMES) + args.warmup + args.steps + WARM_FRAMES,
                "note": "`value` = exactly K steps after W warm-up steps on a device that has been rendering; this = the same on a device "
                        "that has rendered a dozen frames since the process started"},
            "build_id": gh.build_id(),
            "stage_roofline": {
                # 52 N covers the whole sort path of wasm.cpp (key + min/max: 28 N, radix passes: 24 N): over key pass + radix
                # stage (the sort-only leg); the radix stage alone moves 24 N + the 8 N of keys it writes and re-reads
                "sort": (lambda t: {"bytes": b_sort, "ms": t, "GBps": b_sort / (t * 1e-3) / 1e9 if t else 0,
                                    "frac": b_sort / (t * 1e-3) / 1e9 / HBM_PEAK_GBS if t else 0,
                                    "note": "key + min/max pass and radix stage together (gsr_sort)"})(
                    (solo["sort_only"]["ms_key_minmax"] + solo["sort_only"]["ms_quantise_radix"]) if solo else sm["project_key"] + sm["sort"]),
                "radix": {"bytes": 32.0 * N, "ms": sm["sort"], "GBps": 32.0 * N / (sm["sort"] * 1e-3) / 1e9 if sm["sort"] else 0,
                          "frac": 32.0 * N / (sm["sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS if sm["sort"] else 0},
                "project_key": {"bytes": b_proj + 16.0 * N, "ms": sm["project_key"],
                                "GBps": (b_proj + 16.0 * N) / (sm["project_key"] * 1e-3) / 1e9 if sm["project_key"] else 0},
                "bin": {"bytes": b_bin, "ms": sm["bin"], "GBps": b_bin / (sm["bin"] * 1e-3) / 1e9 if sm["bin"] else 0},
                "frame": {"bytes": b_sort + b_proj + b_bin + b_blend, "ms": ms_step,
                          "frac": (b_sort + b_proj + b_bin + b_blend) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            },
            "one_frame_in_flight": solo,
            "band_edges": band_edges_used,
            "device": r.device_info(),
        }
        if world == 1 and args.config == "C3" and not args.timed_only and not args.no_other_configs and not emu:
            out["other_configs"] = {name: other_config(gh, name, local_rank, F) for name in ("C2", "C4")}
            out["other_configs"]["C3_sh"] = sh_config(gh, local_rank, F)
            out["other_configs"]["scene_build"] = scene_build_config(gh, local_rank)
        if c5 is not None:
            out["other_configs"] = {"C5": c5}
        if world == 1 and not args.no_cpu_baseline and not args.timed_only:
            out["cpu_baseline"] = cpu_baseline(gh, cfg, scene.data[:8 * N], scene.positions)
        print(json.dumps(out))
    for rr in reversed(rs):    # (contexts that share rs[0]'s communicator go first)
        rr.dispose()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
