"""N>1 data path rehearsed on one GPU: two processes, each compositing one tile-column band on cuda:0 through the
C ABI, exchange their slabs with the FrameExchange bench.py uses (gloo here, RCCL in the real multi-GPU run) and
must both end up with exactly the frame a single full-width context renders."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "gsplat.js_amd", "py"))
    import torch
    import torch.distributed as dist
    import gsplat_hip as gh
    from gsplat_hip import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = gh.synth.CONFIGS["C1"]
        W, H = cfg["width"], cfg["height"]
        scene = gh.Scene()
        scene.setData(gh.synth.config_rows("C1"))
        cam = gh.orbit_camera(14, width=W, height=H, fx=cfg["fx"])
        x0, x1 = bands.band_edges(W, world)[rank]
        r = gh.HIPRenderer(W, H, device=0, band=(x0, x1))
        r.render(scene, cam)
        fb = torch.from_numpy(r.readPixelsFloat())
        di = r.lastDepthIndex()
        # bench.py's RGBA8 path: pack on the renderer's stream -> all-gather (through host memory under gloo) -> one
        # de-slab kernel on torch's stream; two frames, so the slab-reuse event is exercised too
        link = bands.StreamLink(torch, r, "cuda:0")
        x8 = bands.FrameExchange(dist, torch, W, H, rank, world, torch.device("cuda:0"), dtype=torch.uint8, host_staged=True)
        for _ in range(2):
            r.render_async()
            full8 = x8.exchange_native(r, link)
        torch.cuda.synchronize()
        full8 = full8.cpu().numpy()
        r.dispose()
        full = bands.FrameExchange(dist, torch, W, H, rank, world, "cpu").exchange(fb)
        if rank == 0:
            ref = gh.HIPRenderer(W, H, device=0)
            ref.render(scene, cam)
            want = ref.readPixelsFloat()
            ok = np.array_equal(full.numpy(), want) and np.array_equal(di, ref.lastDepthIndex()) and np.array_equal(full8, ref.readPixels())
            ref.dispose()
            open(out, "w").write("ok" if ok else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_band_render_equals_full_frame(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def _exchange_worker(rank, world, port, out):
    """The LIBRARY's exchange (gsr_allgather_frame_async: pack -> collective -> de-slab, event-ordered against the render
    stream) with world > 1 on one GPU: RCCL refuses two ranks per device, so the collective is injected
    (gsr_comm_init_custom) as a host-staged gloo all-gather; everything around it is the product path.  Unequal
    (cost-balanced) bands, two contexts per rank sharing the group (frames in flight), frames back to back; every rank
    must read the single-context RGBA8 frame byte for byte through gsr_read_frame_rgba8.
    (Reference entry that has to keep working on every rank: renderer.render(scene, camera),
    src/renderers/WebGLRenderer.ts:241-296.)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "gsplat.js_amd", "py"))
    import torch
    import torch.distributed as dist
    import gsplat_hip as gh
    from gsplat_hip import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = gh.synth.CONFIGS["C1"]
        W, H = cfg["width"], cfg["height"]
        scene = gh.Scene()
        scene.setData(gh.synth.config_rows("C1"))
        cams = [gh.orbit_camera(k, width=W, height=H, fx=cfg["fx"]) for k in (3, 38, 71, 104)]
        # cost-balanced, hence unequal, edges from a calibration frame (the same on every rank)
        cal = gh.HIPRenderer(W, H, device=0)
        cal.render(scene, cams[0])
        cost = cal.bin_totals().sum(axis=0) + 0.25 * 32 * H
        edges = bands.balanced_edges(W, world, cost * np.linspace(1.0, 3.0, len(cost)))   # (skewed: the scene is symmetric)
        assert len({b - a for a, b in edges}) > 1, edges
        dev = torch.device("cuda:0")

        def allgather(send, recv, nbytes, stream):
            s = torch.cuda.ExternalStream(stream, device=dev)
            s.synchronize()                                   # the band has been packed into the slab
            mine = torch.as_tensor(bands.DevicePointer(send, (nbytes,), "|u1"), device=dev).cpu()
            every = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(every, mine)
            with torch.cuda.stream(s):
                torch.as_tensor(bands.DevicePointer(recv, (world * nbytes,), "|u1"), device=dev).copy_(every)
            s.synchronize()

        a = gh.HIPRenderer(W, H, device=0)      # (contexts of the same kind as `cal`: throughput contexts composite with the
        b = gh.HIPRenderer(W, H, device=0)      #  other kernel, same pixels to f32 association, RGBA8 to a rounding step)
        a.join_group_custom(rank, world, edges, allgather)
        b.share_group(a)
        for r in (a, b):
            r.render(scene, cams[0])
        frames = []
        for k, cam in enumerate(cams):                         # frames back to back, alternating contexts, no host sync
            r = (a, b)[k % 2]
            r.set_camera(cam)
            r.render_async()
            r.allgather_frame_async()
            if k >= 2:
                frames.append((k, r.read_frame()))            # this context's newest frame
        ok = True
        for k, got in frames:
            cal.render(scene, cams[k])
            ok = ok and np.array_equal(got, cal.readPixels())
        # A list overflow on ONE rank of the group: its frame is not composited, its slab says so, and the flag reaches every
        # rank with the all-gather -- ALL ranks refuse that gathered frame (no rank repeats a collective alone) and then render
        # and gather it again together; the overflowing rank has regrown its lists by then.
        a.sync(); b.sync()
        if rank == world - 1:
            a.set_list_capacity(1024)
        a.set_camera(cams[1])
        a.render_async()
        a.allgather_frame_async()
        refused = False
        try:
            a.read_frame()
        except gh.GsplatError as e:
            refused = "not composited" in str(e)
        ok = ok and refused
        a.set_camera(cams[1])
        a.render_async()
        a.allgather_frame_async()
        got = a.read_frame()
        cal.render(scene, cams[1])
        ok = ok and np.array_equal(got, cal.readPixels())
        if rank == world - 1:
            ok = ok and a.stats()["overflow_frames"] >= 1
        try:
            a.sync()
        except gh.GsplatError as e:          # (nothing was dropped: the one overflowing frame was rendered again)
            ok = False
        # a leader that leaves first detaches its sharer instead of leaving it with a destroyed stream
        a.leave_group()
        try:
            b.allgather_frame_async()
            ok = False
        except gh.GsplatError:
            pass
        b.dispose(); a.dispose(); cal.dispose()
        res = torch.tensor([1 if ok else 0])
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(out, "w").write("ok" if int(res.item()) == 1 else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_in_library_exchange_with_more_than_one_rank(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_exchange_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"
