"""N>1 host path on CPU: two gloo ranks each hold one tile-column band of a frame, exchange slabs with
the same FrameExchange bench.py uses over RCCL, and must both end up with the full frame."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gsplat.js_amd", "py"))
    import torch
    import torch.distributed as dist
    from gsplat_hip import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        full = torch.rand((H, W, 4), generator=g)
        x0, x1 = bands.band_edges(W, world)[rank]
        fb = torch.zeros_like(full)
        fb[:, x0:x1] = full[:, x0:x1]           # what this rank's compositor wrote
        x = bands.FrameExchange(dist, torch, W, H, rank, world, "cpu")
        for _ in range(2):                       # twice: buffers are reused frame after frame
            got = x.exchange(fb)
        ok = bool(torch.equal(got, full))
        # the bench's default: cost-balanced (unequal) bands, RGBA8 slabs
        cost = [1.0 + 50.0 * (abs(k - (W // 32) / 2) < 2) for k in range(-(-W // 32))]
        edges = bands.balanced_edges(W, world, cost)
        full8 = (full * 255).to(torch.uint8)
        a, b = edges[rank]
        fb8 = torch.zeros_like(full8)
        fb8[:, a:b] = full8[:, a:b]
        x8 = bands.FrameExchange(dist, torch, W, H, rank, world, "cpu", edges=edges, dtype=torch.uint8)
        ok = ok and bool(torch.equal(x8.exchange(fb8), full8))
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(out, "w").write("ok" if t.item() == 1.0 else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H,world", [(640, 48, 2), (333, 21, 2), (1920, 16, 3)])
def test_band_exchange_gloo(tmp_path, W, H, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), W, H, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
