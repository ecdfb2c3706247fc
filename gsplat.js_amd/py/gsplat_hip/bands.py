"""Multi-GPU partition of ONE frame: contiguous tile-column
bands, one per rank, then an all-gather of equal-size framebuffer slabs.

The device side only needs gsr_set_band; this module is the host-side
arithmetic and the slab exchange, written against torch.distributed so the same
code runs over RCCL ("nccl" backend on ROCm) on GPUs and over gloo on CPU
tensors in the tests.
"""
BIN_PX = 32  # must match gsr::BIN_PX (band edges are whole compositor bins)


def band_edges(width, world):
    """[(x0, x1)] per rank: whole 32-px bins, equal count per rank, the tail clipped to the image
    (trailing ranks may get an empty band when there are more ranks than bin columns)."""
    nbx = -(-width // BIN_PX)
    per = -(-nbx // world)
    return [(min(q * per * BIN_PX, width), min((q + 1) * per * BIN_PX, width)) for q in range(world)]


def slab_width(width, world):
    nbx = -(-width // BIN_PX)
    return -(-nbx // world) * BIN_PX


class DevicePointer:
    """Expose a raw device pointer to torch (zero copy) through __cuda_array_interface__:
    `torch.as_tensor(DevicePointer(ptr, (H, W, 4)), device="cuda:0")`."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def framebuffer_tensor(torch, renderer, device):
    """The renderer's device framebuffer as a torch tensor [H, W, 4] float32 (no copy)."""
    return torch.as_tensor(DevicePointer(renderer.framebuffer_ptr(), (renderer.height, renderer.width, 4)), device=device)


class FrameExchange:
    """all-gather of per-rank band slabs into every rank's full frame."""

    def __init__(self, dist, torch, width, height, rank, world, device, channels=4):
        self.dist, self.rank, self.world = dist, rank, world
        self.edges = band_edges(width, world)
        sw = slab_width(width, world)
        self.slab = torch.zeros((height, sw, channels), dtype=torch.float32, device=device)
        # concatenated along dim 0 (the layout both NCCL/RCCL and gloo accept for all_gather_into_tensor)
        self._flat = torch.empty((world * height, sw, channels), dtype=torch.float32, device=device)
        self.gathered = self._flat.view(world, height, sw, channels)
        self.full = torch.empty((height, width, channels), dtype=torch.float32, device=device)

    def exchange(self, fb):
        """fb: [H, W, C] tensor whose columns edges[rank] hold this rank's band. Returns the full frame."""
        x0, x1 = self.edges[self.rank]
        if x1 > x0:
            self.slab[:, :x1 - x0].copy_(fb[:, x0:x1])
        self.dist.all_gather_into_tensor(self._flat, self.slab)
        for q, (a, b) in enumerate(self.edges):
            if b > a:
                self.full[:, a:b].copy_(self.gathered[q, :, :b - a])
        return self.full
