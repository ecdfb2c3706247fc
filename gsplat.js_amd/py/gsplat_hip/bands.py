"""Multi-GPU partition of ONE frame: contiguous tile-column
bands, one per rank, then an all-gather of equal-size framebuffer slabs.

The device side only needs gsr_set_band; this module is the host-side
arithmetic and the slab exchange, written against torch.distributed so the same
code runs over RCCL ("nccl" backend on ROCm) on GPUs and over gloo on CPU
tensors in the tests.

xGMI is point-to-point (7 links per GPU), so the exchange is sized for it: by
default the slabs are RGBA8 (8.3 MB per 1080p frame in total, the format a
display consumes) rather than RGBA f32 (33 MB), and there is exactly one
collective per frame.
"""
BIN_PX = 32  # must match gsr::BIN_PX (band edges are whole compositor bins)


def band_edges(width, world):
    """[(x0, x1)] per rank: whole 32-px bins, equal count per rank, the tail clipped to the image
    (trailing ranks may get an empty band when there are more ranks than bin columns)."""
    nbx = -(-width // BIN_PX)
    per = -(-nbx // world)
    return [(min(q * per * BIN_PX, width), min((q + 1) * per * BIN_PX, width)) for q in range(world)]


def balanced_edges(width, world, column_cost):
    """Band edges (whole bins, every rank at least one bin column) that equalise the summed cost.

    column_cost[k]: work estimate of bin column k (e.g. list entries of a calibration frame plus a constant for the
    per-pixel output).  Centre-heavy scenes make equal-width bands badly unbalanced; a frame is as slow as its
    slowest band."""
    nbx = -(-width // BIN_PX)
    cost = [float(c) for c in column_cost]
    assert len(cost) == nbx
    if world >= nbx:
        cuts = list(range(nbx + 1)) + [nbx] * (world - nbx)
    else:
        prefix = [0.0]
        for c in cost:
            prefix.append(prefix[-1] + c)
        cuts = [0]
        for q in range(1, world):
            target = prefix[-1] * q / world
            lo, hi = cuts[-1] + 1, nbx - (world - q)   # leave at least one column for every later rank
            cuts.append(min(range(lo, hi + 1), key=lambda k: abs(prefix[k] - target)))
        cuts.append(nbx)
    return [(min(a * BIN_PX, width), min(b * BIN_PX, width)) for a, b in zip(cuts[:-1], cuts[1:])]


def slab_width(width, world, edges=None):
    if edges is not None:
        return max(BIN_PX, max(b - a for a, b in edges))
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


def framebuffer8_tensor(torch, renderer, device):
    """The renderer's RGBA8 device image (filled by convert_rgba8_async) as a torch tensor [H, W, 4] uint8."""
    return torch.as_tensor(DevicePointer(renderer.framebuffer8_ptr(), (renderer.height, renderer.width, 4), "|u1"), device=device)


class StreamLink:
    """Orders the library's HIP stream and torch's current stream on the device, without host round trips."""

    def __init__(self, torch, renderer, device):
        self.torch = torch
        self.renderer = renderer
        self.ext = torch.cuda.ExternalStream(renderer.stream_handle(), device=device)

    def torch_waits_for_renderer(self):
        # one event record + one stream wait inside the library (a third of the cost of the torch calls)
        self.renderer.stream_order(self.torch.cuda.current_stream().cuda_stream, False)

    def renderer_waits_for_torch(self):
        self.renderer.stream_order(self.torch.cuda.current_stream().cuda_stream, True)

    def renderer_waits_for_event(self, event):
        """Narrower than renderer_waits_for_torch: the renderer's stream waits only for `event` (e.g. "my band has been
        copied out of the framebuffer"), so the collective that follows on torch's side overlaps the next frame."""
        self.ext.wait_event(event)


class FrameExchange:
    """all-gather of per-rank band slabs into every rank's full frame.

    Two ways to use it: `exchange(fb)` works on any torch tensor (CPU/gloo in the tests, f32 or u8);
    `exchange_native(renderer, link)` is the GPU RGBA8 path: the library packs the band straight into the slab
    (gsr_pack_band_rgba8_async) and de-slabs the gathered buffer with one kernel (gsr_unpack_slabs_rgba8_async),
    so a frame costs one collective and no torch copy kernels."""

    def __init__(self, dist, torch, width, height, rank, world, device, channels=4, edges=None, dtype=None, host_staged=False):
        self.dist, self.rank, self.world = dist, rank, world
        # host_staged: rehearsal of the N>1 path on a box without RCCL peers (gloo has no device all-gather): the slab
        # takes a round trip through host memory around the collective; everything else is the production path
        self.host_staged = host_staged
        self.edges = list(edges) if edges is not None else band_edges(width, world)
        dtype = dtype or torch.float32
        sw = slab_width(width, world, self.edges)
        self.slab = torch.zeros((height, sw, channels), dtype=dtype, device=device)
        # concatenated along dim 0 (the layout both NCCL/RCCL and gloo accept for all_gather_into_tensor)
        self._flat = torch.empty((world * height, sw, channels), dtype=dtype, device=device)
        self.gathered = self._flat.view(world, height, sw, channels)
        self.full = torch.empty((height, width, channels), dtype=dtype, device=device)
        # recorded on the current stream right after the band has been copied into the slab (device tensors only)
        self.copied = torch.cuda.Event() if str(device).startswith("cuda") else None
        self._exchanged = False
        self._edge_arrays = None          # ctypes copies of the edges for the de-slab call (built once)
        self._ptrs = None

    def exchange_native(self, renderer, link):
        """RGBA8 exchange of the frame `renderer` has enqueued.  Device-side ordering only:
        pack on the renderer's stream -> torch's stream waits -> all-gather -> de-slab on torch's stream.
        The slab is reused by the next call, so the next pack waits for what torch's stream holds by then (this all-gather)."""
        torch = link.torch
        sw = self.slab.shape[1]
        if self._edge_arrays is None:
            from . import edge_arrays
            self._edge_arrays = edge_arrays(self.edges)
            self._ptrs = (self.slab.data_ptr(), self._flat.data_ptr(), self.full.data_ptr())
        if renderer.overflow_pending():
            # never ship a band the compositor did not draw (a frame whose lists did not fit publishes no work):
            # sync regrows the lists and renders the frame again; frames lost earlier stay counted in stats()
            try:
                renderer.sync()
            except Exception as e:
                if "not composited" not in str(e):
                    raise
        if self._exchanged:
            link.renderer_waits_for_torch()      # the previous collective (and de-slab) has read the slab
        self._exchanged = True
        renderer.pack_band_rgba8_async(self._ptrs[0], sw)
        link.torch_waits_for_renderer()
        if self.host_staged:
            host_slab = self.slab.cpu()
            host_flat = torch.empty(self._flat.shape, dtype=self._flat.dtype)
            self.dist.all_gather_into_tensor(host_flat, host_slab)
            self._flat.copy_(host_flat)
        else:
            self.dist.all_gather_into_tensor(self._flat, self.slab)
        renderer.unpack_slabs_rgba8_async(self._ptrs[1], self._ptrs[2], sw, self._edge_arrays,
                                          torch.cuda.current_stream().cuda_stream)
        return self.full

    def exchange(self, fb):
        """fb: [H, W, C] tensor whose columns edges[rank] hold this rank's band. Returns the full frame."""
        x0, x1 = self.edges[self.rank]
        if x1 > x0:
            self.slab[:, :x1 - x0].copy_(fb[:, x0:x1])
        if self.copied is not None:
            self.copied.record()        # from here on the producer may overwrite fb
        if self.host_staged and self.slab.is_cuda:   # rehearsal under gloo: no device all-gather
            host_flat = self._flat.cpu()
            self.dist.all_gather_into_tensor(host_flat, self.slab.cpu())
            self._flat.copy_(host_flat)
        else:
            self.dist.all_gather_into_tensor(self._flat, self.slab)
        for q, (a, b) in enumerate(self.edges):
            if b > a:
                self.full[:, a:b].copy_(self.gathered[q, :, :b - a])
        return self.full
