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
