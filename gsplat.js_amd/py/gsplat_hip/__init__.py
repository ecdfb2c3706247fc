"""Python (ctypes) host harness over the C ABI of libgsplat_hip.so.

This is plumbing for the tests and bench.py: it mirrors the reference's
interface for the hot path -- `Scene.setData`, `Camera.update`,
`renderer.render(scene, camera)` (src/core/Scene.ts:58-180,
src/cameras/Camera.ts:81-92, src/renderers/WebGLRenderer.ts:241-296) -- and
calls the HIP library for everything the GPU does.  There is no CPU fallback:
if the library or a GPU is missing, construction raises.
The Node/JavaScript host (the reference's own language) lives in ../../js.
"""
import ctypes
import os

import numpy as np

from .camera import Camera, orbit_camera, orbit_pose  # noqa: F401
from . import synth  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libgsplat_hip.so"))

GSR_FLAG_TIMING = 1
GSR_FLAG_THROUGHPUT = 2


class GsrOptions(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("early_out_eps", ctypes.c_float), ("band_x0", ctypes.c_int32), ("band_x1", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class GsrTimings(ctypes.Structure):
    _fields_ = [("ms_project_key", ctypes.c_float), ("ms_sort", ctypes.c_float), ("ms_bin", ctypes.c_float),
                ("ms_blend", ctypes.c_float), ("ms_combine", ctypes.c_float), ("ms_total", ctypes.c_float), ("visible", ctypes.c_uint64),
                ("bin_entries", ctypes.c_uint64), ("tile_entries", ctypes.c_uint64), ("n", ctypes.c_uint32), ("frames", ctypes.c_uint32),
                ("sum_ms_project_key", ctypes.c_double), ("sum_ms_sort", ctypes.c_double),
                ("sum_ms_bin", ctypes.c_double), ("sum_ms_blend", ctypes.c_double), ("sum_ms_combine", ctypes.c_double),
                ("sum_ms_total", ctypes.c_double),
                ("sum_visible", ctypes.c_uint64), ("sum_bin_entries", ctypes.c_uint64),
                ("sum_tile_entries", ctypes.c_uint64), ("sum_frames", ctypes.c_uint64),
                ("overflow_frames", ctypes.c_uint64), ("dropped_frames", ctypes.c_uint64)]


def edge_arrays(edges):
    """[(x0, x1)] per rank -> (c_int32[world], c_int32[world]) for gsr_unpack_slabs_rgba8_async."""
    world = len(edges)
    return ((ctypes.c_int32 * world)(*[int(a) for a, _ in edges]), (ctypes.c_int32 * world)(*[int(b) for _, b in edges]))


class GsplatError(RuntimeError):
    pass


_lib = None

# every symbol include/gsplat_hip.h declares
EXPORTS = [
    "gsr_create", "gsr_destroy", "gsr_last_error", "gsr_set_scene", "gsr_set_scene_sh", "gsr_read_sh_colors", "gsr_set_depth_fade", "gsr_resize",
    "gsr_set_scene_rows", "gsr_scene_translate", "gsr_scene_rotate", "gsr_scene_scale", "gsr_scene_limit_box", "gsr_read_scene", "gsr_set_band", "gsr_set_camera",
    "gsr_sort", "gsr_render", "gsr_render_async", "gsr_sync", "gsr_read_depth_index", "gsr_read_pixels_rgba32f",
    "gsr_read_pixels_rgba8", "gsr_get_timings", "gsr_reset_timings", "gsr_set_timing_interval", "gsr_read_keys", "gsr_read_records",
    "gsr_read_bin_totals", "gsr_read_bin_lists", "gsr_convert_rgba8_async", "gsr_framebuffer8_device_ptr",
    "gsr_pack_band_rgba8_async", "gsr_unpack_slabs_rgba8_async",
    "gsr_framebuffer_device_ptr", "gsr_stream_handle", "gsr_stream_order", "gsr_device_info", "gsplat_sort_host",
    "gsr_overflow_pending", "gsr_set_list_capacity", "gsr_scene_count", "gsr_build_id",
    "gsr_comm_unique_id", "gsr_comm_init", "gsr_comm_destroy", "gsr_allgather_frame_async", "gsr_read_frame_rgba8",
    "gsr_frame8_device_ptr", "gsr_comm_stream_handle", "gsr_read_work_items", "gsr_comm_share", "gsr_comm_init_custom",
]
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
GSR_COMM_ID_BYTES = 128


def _assert_one_hip_runtime():
    """The library and torch must share ONE libamdhip64 (streams, events and device pointers cross between them).
    Which copy that is depends on load order (see load_library); two mapped copies mean two runtimes in the process,
    which fails in obscure ways later, so fail here with the reason."""
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise GsplatError("two HIP runtimes are mapped into this process (%s): import torch before gsplat_hip, or set "
                          "GSPLAT_HIP_NO_TORCH=1 in a process that never uses torch" % ", ".join(sorted(paths)))


def load_library(path=None):
    """Load libgsplat_hip.so and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GSPLAT_HIP_LIB") or LIB_PATH   # GSPLAT_HIP_LIB: experiment builds
    if not os.path.exists(p):
        raise GsplatError("libgsplat_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system one
    # this library links against).  If torch is imported first, the loader gives this library torch's copy and the two
    # share streams, events and memory (the N>1 exchange relies on that); loaded the other way round, the process ends
    # up with two runtimes and torch cannot see the GPU any more.  So: when torch is installed, import it before the
    # library (GSPLAT_HIP_NO_TORCH=1 skips this for hosts that never touch torch).
    import sys
    if "torch" not in sys.modules and os.environ.get("GSPLAT_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(p)
    _assert_one_hip_runtime()
    vp = ctypes.c_void_p
    L.gsr_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(GsrOptions)]
    L.gsr_destroy.argtypes = [vp]
    L.gsr_last_error.argtypes = [vp]
    L.gsr_last_error.restype = ctypes.c_char_p
    L.gsr_set_scene.argtypes = [vp, vp, vp, ctypes.c_uint32]
    L.gsr_set_scene_sh.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp]
    L.gsr_read_sh_colors.argtypes = [vp, vp]
    L.gsr_set_depth_fade.argtypes = [vp, ctypes.c_int32, ctypes.c_float]
    L.gsr_set_scene_rows.argtypes = [vp, vp, ctypes.c_uint32]
    for name in ("gsr_scene_translate", "gsr_scene_rotate", "gsr_scene_scale"):
        getattr(L, name).argtypes = [vp, vp]
    L.gsr_scene_limit_box.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint32)]
    L.gsr_read_scene.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_uint32)]
    L.gsr_resize.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    L.gsr_set_band.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    L.gsr_set_camera.argtypes = [vp, vp, vp, vp, ctypes.c_float, ctypes.c_float]
    for name in ("gsr_sort", "gsr_render", "gsr_render_async", "gsr_sync", "gsr_reset_timings"):
        getattr(L, name).argtypes = [vp]
    L.gsr_read_depth_index.argtypes = [vp, vp]
    L.gsr_read_pixels_rgba32f.argtypes = [vp, vp]
    L.gsr_read_pixels_rgba8.argtypes = [vp, vp]
    L.gsr_get_timings.argtypes = [vp, ctypes.POINTER(GsrTimings)]
    L.gsr_set_timing_interval.argtypes = [vp, ctypes.c_uint32]
    L.gsr_read_keys.argtypes = [vp, vp, vp]
    L.gsr_read_records.argtypes = [vp, vp, vp]
    L.gsr_read_bin_totals.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.gsr_read_bin_lists.argtypes = [vp, vp, vp, ctypes.c_uint64]
    L.gsr_convert_rgba8_async.argtypes = [vp]
    L.gsr_read_work_items.argtypes = [vp, vp]
    L.gsr_stream_order.argtypes = [vp, vp, ctypes.c_int32]
    L.gsr_pack_band_rgba8_async.argtypes = [vp, vp, ctypes.c_int32]
    L.gsr_unpack_slabs_rgba8_async.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                               ctypes.POINTER(ctypes.c_int32), vp]
    L.gsr_framebuffer8_device_ptr.argtypes = [vp]
    L.gsr_framebuffer8_device_ptr.restype = vp
    L.gsr_framebuffer_device_ptr.argtypes = [vp]
    L.gsr_framebuffer_device_ptr.restype = vp
    L.gsr_stream_handle.argtypes = [vp]
    L.gsr_stream_handle.restype = vp
    L.gsr_device_info.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                  ctypes.POINTER(ctypes.c_int32)]
    L.gsplat_sort_host.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
    L.gsplat_sort_host.restype = None
    L.gsr_overflow_pending.argtypes = [vp]
    L.gsr_set_list_capacity.argtypes = [vp, ctypes.c_uint32]
    L.gsr_scene_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.gsr_build_id.argtypes = []
    L.gsr_build_id.restype = ctypes.c_char_p
    L.gsr_comm_unique_id.argtypes = [vp]
    L.gsr_comm_init.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.gsr_comm_share.argtypes = [vp, vp]
    L.gsr_comm_init_custom.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                       ALLGATHER_FN, vp]
    L.gsr_comm_destroy.argtypes = [vp]
    L.gsr_allgather_frame_async.argtypes = [vp]
    L.gsr_read_frame_rgba8.argtypes = [vp, vp]
    L.gsr_frame8_device_ptr.argtypes = [vp]
    L.gsr_frame8_device_ptr.restype = vp
    L.gsr_comm_stream_handle.argtypes = [vp]
    L.gsr_comm_stream_handle.restype = vp
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("gsplat_sort_host",):
            fn.restype = ctypes.c_int
    if path is None:
        _lib = L
    return L


# ---------------------------------------------------------------------------
# Scene: mirror of src/core/Scene.ts (setData only: the producer of the buffers
# the hot path consumes), vectorised in numpy float64 = JS number arithmetic.
# ---------------------------------------------------------------------------
def _float_to_half(x64):
    """src/utils.ts:16-43 (truncating; JS `>>` shift count taken modulo 32)."""
    with np.errstate(over="ignore", invalid="ignore"):
        f = np.asarray(x64, dtype=np.float64).astype(np.float32).view(np.int32).astype(np.int64)
    sign = (f >> 31) & 1
    exp = (f >> 23) & 0xFF
    frac = f & 0x007FFFFF
    sub = (exp > 0) & (exp < 113)
    shift = np.where(sub, (113 - exp) & 31, 0)
    frac_sub = (frac | 0x00800000) >> shift
    carry = sub & ((frac_sub & 0x01000000) != 0)
    new_exp = np.where(exp == 0, 0, np.where(exp < 113, 0, np.where(exp < 142, exp - 112, 31)))
    new_exp = np.where(carry, 1, new_exp)
    frac_out = np.where(sub, frac_sub, frac)
    frac_out = np.where(carry | (exp >= 142), 0, frac_out)
    return ((sign << 15) | (new_exp << 10) | (frac_out >> 13)).astype(np.uint32)


def pack_half2x16(x, y):
    """src/utils.ts:46-48."""
    return (_float_to_half(x) | (_float_to_half(y) << np.uint32(16))).astype(np.uint32)


class Scene:
    RowLength = 32  # src/core/Scene.ts:9

    def __init__(self):
        self._listeners = {}
        self.data = np.zeros(0, dtype=np.uint32)
        self.positions = np.zeros(0, dtype=np.float32)
        self.vertexCount = 0
        self.width = 2048
        self.height = 0
        self.shHeight = 0
        self.shs_rgb = [np.zeros(0, dtype=np.uint32) for _ in range(3)]
        self.bandsIndices = np.array([-1, -1, -1], dtype=np.int32)

    # EventDispatcher surface used by the renderer (src/core/EventDispatcher.ts)
    def addEventListener(self, kind, fn):
        self._listeners.setdefault(kind, []).append(fn)

    def removeEventListener(self, kind, fn):
        if fn in self._listeners.get(kind, []):
            self._listeners[kind].remove(fn)

    def dispatchEvent(self, kind):
        for fn in list(self._listeners.get(kind, [])):
            fn({"type": kind})

    def setData(self, rows, shs=None):
        """Scene.ts:58-180.  shs: 48 floats per SH-carrying splat (set bandsIndices first, like the loader does)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1)
        if rows.size % self.RowLength:
            raise ValueError("data length must be a multiple of %d" % self.RowLength)
        n = rows.size // self.RowLength
        self.vertexCount = n
        self.height = -(-(2 * n) // self.width)
        r = rows.reshape(n, 32)
        f = r[:, :24].copy().view(np.float32).reshape(n, 6)
        self.positions = f[:, 0:3].copy().reshape(-1)
        data = np.zeros((self.width * self.height * 4) if n else 0, dtype=np.uint32)
        d = data[:8 * n].reshape(n, 8)
        d[:, 0:3] = f[:, 0:3].copy().view(np.uint32)
        d[:, 7] = r[:, 24:28].copy().view(np.uint32).reshape(n)
        rot = (r[:, 28:32].astype(np.float64) - 128.0) / 128.0
        qx, qy, qz, qw = rot[:, 1], rot[:, 2], rot[:, 3], -rot[:, 0]
        R = [1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw,
             2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw,
             2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy]
        s = f[:, 3:6].astype(np.float64)
        z = np.zeros(n)
        a = [s[:, 0], z, z, z, s[:, 1], z, z, z, s[:, 2]]
        b = R
        M = [b[0] * a[0] + b[3] * a[1] + b[6] * a[2], b[1] * a[0] + b[4] * a[1] + b[7] * a[2], b[2] * a[0] + b[5] * a[1] + b[8] * a[2],
             b[0] * a[3] + b[3] * a[4] + b[6] * a[5], b[1] * a[3] + b[4] * a[4] + b[7] * a[5], b[2] * a[3] + b[5] * a[4] + b[8] * a[5],
             b[0] * a[6] + b[3] * a[7] + b[6] * a[8], b[1] * a[6] + b[4] * a[7] + b[7] * a[8], b[2] * a[6] + b[5] * a[7] + b[8] * a[8]]
        sg = [M[0] * M[0] + M[3] * M[3] + M[6] * M[6], M[0] * M[1] + M[3] * M[4] + M[6] * M[7],
              M[0] * M[2] + M[3] * M[5] + M[6] * M[8], M[1] * M[1] + M[4] * M[4] + M[7] * M[7],
              M[1] * M[2] + M[4] * M[5] + M[7] * M[8], M[2] * M[2] + M[5] * M[5] + M[8] * M[8]]
        d[:, 4] = pack_half2x16(4 * sg[0], 4 * sg[1])
        d[:, 5] = pack_half2x16(4 * sg[2], 4 * sg[3])
        d[:, 6] = pack_half2x16(4 * sg[4], 4 * sg[5])
        self.data = data
        if shs is not None:   # Scene.ts:83-124: three half textures, one per colour channel
            shs = np.ascontiguousarray(shs, dtype=np.float32).reshape(-1)
            count = n - (int(self.bandsIndices[0]) + 1)
            self.shHeight = -(-(2 * count) // self.width)
            c = shs[:count * 48].reshape(count, 8, 2, 3).astype(np.float64)   # (splat, word, half, channel)
            self.shs_rgb = []
            for ch in range(3):
                tex = np.zeros(self.width * self.shHeight * 4, dtype=np.uint32)
                tex[:8 * count] = pack_half2x16(c[:, :, 0, ch].reshape(-1), c[:, :, 1, ch].reshape(-1))
                self.shs_rgb.append(tex)
        else:
            self.shHeight = 0
        self.dispatchEvent("change")


# ---------------------------------------------------------------------------
# HIPRenderer: the drop-in for WebGLRenderer's render path
# ---------------------------------------------------------------------------
class HIPRenderer:
    """renderer.render(scene, camera) on an MI355X (WebGLRenderer.ts:241-296)."""

    def __init__(self, width=1920, height=1080, device=0, early_out_eps=0.0, band=None, timing=False, lib_path=None,
                 throughput=False):
        self._L = load_library(lib_path)
        self._ctx = ctypes.c_void_p()
        opt = GsrOptions(device, width, height, early_out_eps, band[0] if band else 0, band[1] if band else 0,
                         (GSR_FLAG_TIMING if timing else 0) | (GSR_FLAG_THROUGHPUT if throughput else 0))
        rc = self._L.gsr_create(ctypes.byref(self._ctx), ctypes.byref(opt))
        if rc:
            raise GsplatError("gsr_create failed (%d): %s" % (rc, self._L.gsr_last_error(None).decode()))
        self.width, self.height = width, height
        self._scene = None
        self._camera = None
        self._n = 0
        self._on_change = lambda _e: self._upload(self._scene)

    # -- helpers --
    def _check(self, rc):
        if rc:
            raise GsplatError("libgsplat_hip error %d: %s" % (rc, self._L.gsr_last_error(self._ctx).decode()))

    def _upload(self, scene):
        data = np.ascontiguousarray(scene.data, dtype=np.uint32)
        pos = np.ascontiguousarray(scene.positions, dtype=np.float32)
        self._check(self._L.gsr_set_scene(self._ctx, data.ctypes.data, pos.ctypes.data, scene.vertexCount))
        self._n = scene.vertexCount
        if getattr(scene, "shHeight", 0):   # WebGLRenderer.ts:202-211: SH textures + u_bandIndex only when the scene has them
            self.set_sh(scene.shs_rgb, scene.bandsIndices)

    # -- reference surface --
    def setSize(self, width, height):
        self._check(self._L.gsr_resize(self._ctx, width, height))
        self.width, self.height = width, height

    def set_band(self, x0, x1):
        self._check(self._L.gsr_set_band(self._ctx, x0, x1))

    def set_raw_scene(self, data, positions):
        """Upload Scene.data / Scene.positions arrays directly (no Scene object)."""
        data = np.ascontiguousarray(data, dtype=np.uint32)
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        n = pos.size // 3
        self._check(self._L.gsr_set_scene(self._ctx, data.ctypes.data, pos.ctypes.data, n))
        self._n = n
        self._scene = None

    def set_sh(self, shs_rgb, bands_indices):
        band = np.ascontiguousarray(bands_indices, dtype=np.int32)
        count = self._n - (int(band[0]) + 1)
        tex = [np.ascontiguousarray(t, dtype=np.uint32) for t in shs_rgb]
        self._check(self._L.gsr_set_scene_sh(self._ctx, tex[0].ctypes.data, tex[1].ctypes.data, tex[2].ctypes.data, count,
                                             band.ctypes.data))

    def read_sh_colors(self):
        out = np.empty((self._n, 4), dtype=np.float32)
        self._check(self._L.gsr_read_sh_colors(self._ctx, out.ctypes.data))
        return out

    # -- on-device scene build and transforms (Scene.ts:58-366 as kernels) --
    def set_scene_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1)
        self._check(self._L.gsr_set_scene_rows(self._ctx, rows.ctypes.data, rows.size // 32))
        self._n = rows.size // 32
        self._scene = None

    def scene_translate(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        self._check(self._L.gsr_scene_translate(self._ctx, t.ctypes.data))

    def scene_rotate(self, q_xyzw):
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        self._check(self._L.gsr_scene_rotate(self._ctx, q.ctypes.data))

    def scene_scale(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self._check(self._L.gsr_scene_scale(self._ctx, s.ctypes.data))

    def scene_limit_box(self, box):
        box = np.ascontiguousarray(box, dtype=np.float64)
        n = ctypes.c_uint32(0)
        self._check(self._L.gsr_scene_limit_box(self._ctx, box.ctypes.data, ctypes.byref(n)))
        self._n = n.value
        return n.value

    def read_scene(self, with_rows=True):
        """(data u32[8n], positions f32[3n], rotations f32[4n] | None, scales f32[3n] | None)"""
        n = self._n
        data = np.zeros(8 * n, dtype=np.uint32)
        pos = np.zeros(3 * n, dtype=np.float32)
        rot = np.zeros(4 * n, dtype=np.float32) if with_rows else None
        scl = np.zeros(3 * n, dtype=np.float32) if with_rows else None
        cnt = ctypes.c_uint32(0)
        self._check(self._L.gsr_read_scene(self._ctx, data.ctypes.data, pos.ctypes.data, rot.ctypes.data if with_rows else None,
                                           scl.ctypes.data if with_rows else None, ctypes.byref(cnt)))
        assert cnt.value == n
        return data, pos, rot, scl

    def set_depth_fade(self, use, value):
        """u_useDepthFade / u_depthFade of FadeInPass."""
        self._check(self._L.gsr_set_depth_fade(self._ctx, 1 if use else 0, float(value)))

    def set_camera(self, camera):
        camera.update(self.width, self.height)
        v, p, vp = camera.f32()
        self._check(self._L.gsr_set_camera(self._ctx, v.ctypes.data, p.ctypes.data, vp.ctypes.data, camera.fx, camera.fy))
        self._camera = camera

    def set_camera_arrays(self, view, proj, view_proj, fx, fy):
        """Pre-rounded float32[16] matrices (what `new Float32Array(m.buffer)` yields)."""
        self._check(self._L.gsr_set_camera(self._ctx, view.ctypes.data, proj.ctypes.data, view_proj.ctypes.data, fx, fy))

    def render(self, scene, camera, sync=True):
        if scene is not None and scene is not self._scene:
            if self._scene is not None:
                self._scene.removeEventListener("change", self._on_change)
            self._scene = scene
            scene.addEventListener("change", self._on_change)
            self._upload(scene)
        self.set_camera(camera)
        self._check(self._L.gsr_render(self._ctx) if sync else self._L.gsr_render_async(self._ctx))

    def render_async(self):
        self._check(self._L.gsr_render_async(self._ctx))

    def sync(self):
        """Wait for the enqueued frames.  Raises GsplatError (code -5) once if asynchronous frames were lost to a list
        overflow; the lists have been regrown by then and the renderer stays usable."""
        self._check(self._L.gsr_sync(self._ctx))

    def overflow_pending(self):
        """True while the device has reported a list overflow that the host has not handled (no copy, no sync)."""
        return bool(self._L.gsr_overflow_pending(self._ctx))

    def set_list_capacity(self, entries):
        """Tuning/test hook: capacity of the bin-list buffer in entries (call after the scene is uploaded)."""
        self._check(self._L.gsr_set_list_capacity(self._ctx, int(entries)))

    def scene_count(self):
        n = ctypes.c_uint32(0)
        self._check(self._L.gsr_scene_count(self._ctx, ctypes.byref(n)))
        return n.value

    def sort(self, camera=None):
        if camera is not None:
            self.set_camera(camera)
        self._check(self._L.gsr_sort(self._ctx))

    def dispose(self):
        if self._ctx:
            self._L.gsr_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    # -- results --
    def lastDepthIndex(self):
        out = np.empty(self._n, dtype=np.uint32)
        self._check(self._L.gsr_read_depth_index(self._ctx, out.ctypes.data))
        return out

    def readPixelsFloat(self, out=None):
        """premultiplied RGBA float32 [H, W, 4], row 0 = top; `out` (C-contiguous, that shape and dtype) is filled and returned"""
        if out is None:
            out = np.empty((self.height, self.width, 4), dtype=np.float32)
        elif out.dtype != np.float32 or out.size != self.height * self.width * 4 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of height*width*4 elements")
        self._check(self._L.gsr_read_pixels_rgba32f(self._ctx, out.ctypes.data))
        return out

    def readPixels(self, out=None):
        """RGBA8 [H, W, 4], row 0 = top; `out` (C-contiguous uint8, height*width*4 elements) is filled and returned"""
        if out is None:
            out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != self.height * self.width * 4 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 array of height*width*4 elements")
        self._check(self._L.gsr_read_pixels_rgba8(self._ctx, out.ctypes.data))
        return out

    def read_keys(self):
        keys = np.empty(self._n, dtype=np.uint32)
        mm = np.zeros(2, dtype=np.int32)
        self._check(self._L.gsr_read_keys(self._ctx, keys.ctypes.data, mm.ctypes.data))
        return keys, (int(mm[0]), int(mm[1]))

    def read_records(self):
        rec = np.empty((self._n, 8), dtype=np.float32)
        bbox = np.empty((self._n, 4), dtype=np.int32)
        self._check(self._L.gsr_read_records(self._ctx, rec.ctypes.data, bbox.ctypes.data))
        return rec, bbox

    def stats(self):
        t = GsrTimings()
        self._check(self._L.gsr_get_timings(self._ctx, ctypes.byref(t)))
        return {k: getattr(t, k) for k, _ in GsrTimings._fields_}

    def reset_stats(self):
        self._check(self._L.gsr_reset_timings(self._ctx))

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cus, clk = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._L.gsr_device_info(self._ctx, name, 256, ctypes.byref(cus), ctypes.byref(clk)))
        return {"name": name.value.decode(), "compute_units": cus.value, "clock_khz": clk.value}

    def bin_totals(self):
        """Entries per 32x32 bin of the last frame, shape [nby, nbx] (this context's band)."""
        nbx_all, nby_all = -(-self.width // 32), -(-self.height // 32)
        out = np.zeros(nbx_all * nby_all, dtype=np.uint32)
        nbx, nby = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._L.gsr_read_bin_totals(self._ctx, out.ctypes.data, ctypes.byref(nbx), ctypes.byref(nby)))
        return out[:nbx.value * nby.value].reshape(nby.value, nbx.value)

    def bin_lists(self):
        """The last frame's bin lists: (starts [bins + 1], list [entries]) -- splat indices, front to back inside each 32x32 bin."""
        nbins = self.bin_totals().size
        starts = np.zeros(nbins + 1, dtype=np.uint32)
        self._check(self._L.gsr_read_bin_lists(self._ctx, starts.ctypes.data, None, 0))
        lst = np.zeros(max(int(starts[-1]), 1), dtype=np.uint32)
        self._check(self._L.gsr_read_bin_lists(self._ctx, starts.ctypes.data, lst.ctypes.data, lst.size))
        return starts, lst[:int(starts[-1])]

    def work_items(self):
        """How the last frame's bin lists were cut for the compositor: entries per segment, work items, the compositor's waves
        per 16x16 tile (which of its two kernels ran), bins.  ("speculative" is always False: the option is gone.)"""
        out = np.zeros(5, dtype=np.uint32)
        self._check(self._L.gsr_read_work_items(self._ctx, out.ctypes.data))
        return {"seg_len": int(out[0]), "items": int(out[1]), "speculative": bool(out[2]), "waves_per_tile": int(out[3]), "bins": int(out[4])}

    def set_timing_interval(self, every):
        """Record stage events only on every `every`-th frame (they cost command-processor time on short frames)."""
        self._check(self._L.gsr_set_timing_interval(self._ctx, every))

    def convert_rgba8_async(self):
        self._check(self._L.gsr_convert_rgba8_async(self._ctx))

    def pack_band_rgba8_async(self, slab_ptr, slab_width_px):
        """This context's band as RGBA8 into the all-gather slab (device pointer), on the renderer's stream."""
        self._check(self._L.gsr_pack_band_rgba8_async(self._ctx, ctypes.c_void_p(slab_ptr), slab_width_px))

    def unpack_slabs_rgba8_async(self, gathered_ptr, image_ptr, slab_width_px, edges, stream_handle):
        """Gathered slabs [world][H][slab_w] -> row-major image, on `stream_handle` (the collective's stream).
        `edges`: [(x0, x1)] per rank, or the pair of ctypes arrays `edge_arrays(edges)` returns (per-frame callers)."""
        x0, x1 = edges if isinstance(edges, tuple) and not isinstance(edges[0], (tuple, list)) else edge_arrays(edges)
        self._check(self._L.gsr_unpack_slabs_rgba8_async(self._ctx, ctypes.c_void_p(gathered_ptr), ctypes.c_void_p(image_ptr),
                                                         slab_width_px, len(x0), x0, x1, ctypes.c_void_p(stream_handle)))

    def framebuffer8_ptr(self):
        return self._L.gsr_framebuffer8_device_ptr(self._ctx)

    def framebuffer_ptr(self):
        return self._L.gsr_framebuffer_device_ptr(self._ctx)

    def stream_order(self, other_stream_handle, ctx_waits):
        """Device-side ordering with another stream (see gsr_stream_order)."""
        self._check(self._L.gsr_stream_order(self._ctx, ctypes.c_void_p(other_stream_handle), 1 if ctx_waits else 0))

    def stream_handle(self):
        return self._L.gsr_stream_handle(self._ctx)

    # -- multi-GPU frame exchange inside the library (RCCL all-gather; see include/gsplat_hip.h) --
    def join_group(self, comm_id, rank, world, edges):
        """Collective: every rank calls it with the same 128-byte id (new_group_id() on rank 0, handed round by the
        host) and the same band edges [(x0, x1)] per rank.  Afterwards render_async() + allgather_frame_async()
        leave the whole RGBA8 frame on every rank (read_frame())."""
        cid = (ctypes.c_uint8 * GSR_COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        x0, x1 = edge_arrays(edges)
        self._check(self._L.gsr_comm_init(self._ctx, cid, rank, world, x0, x1))

    def share_group(self, leader):
        """This context (another frame in flight of the same rank) uses `leader`'s communicator and exchange stream."""
        self._check(self._L.gsr_comm_share(self._ctx, leader._ctx))

    def join_group_custom(self, rank, world, edges, allgather):
        """Test hook (gsr_comm_init_custom): `allgather(send_ptr, recv_ptr, bytes_per_rank, stream)` replaces ncclAllGather."""
        x0, x1 = edge_arrays(edges)

        def _cb(user, send, recv, nbytes, stream):
            try:
                allgather(send, recv, int(nbytes), stream)
                return 0
            except Exception:      # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._allgather_cb = ALLGATHER_FN(_cb)   # keep the trampoline alive as long as the context
        self._check(self._L.gsr_comm_init_custom(self._ctx, rank, world, x0, x1, self._allgather_cb, None))

    def leave_group(self):
        self._check(self._L.gsr_comm_destroy(self._ctx))

    def allgather_frame_async(self):
        self._check(self._L.gsr_allgather_frame_async(self._ctx))

    def read_frame(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._L.gsr_read_frame_rgba8(self._ctx, out.ctypes.data))
        return out

    def frame8_ptr(self):
        return self._L.gsr_frame8_device_ptr(self._ctx)


def new_group_id():
    """128 bytes identifying a new RCCL communicator (rank 0 creates it; the host distributes it)."""
    buf = (ctypes.c_uint8 * GSR_COMM_ID_BYTES)()
    L = load_library()
    rc = L.gsr_comm_unique_id(buf)
    if rc:
        raise GsplatError("gsr_comm_unique_id failed (%d): %s" % (rc, L.gsr_last_error(None).decode()))
    return bytes(buf)


def build_id():
    """Hash of the kernel sources the loaded library was built from."""
    return load_library().gsr_build_id().decode()


WebGLRenderer = HIPRenderer  # the name callers of the reference use (src/index.ts:5)
