"""ctypes front-end of the CPU oracle (oracle/oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY -- imported by tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); never by the product package.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
i32p = ctypes.POINTER(ctypes.c_int32)
f32p = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.run(["make", "-C", HERE, "--no-print-directory"], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_sort.argtypes = [f32p, ctypes.c_uint32, f32p, u32p, u32p, i32p]
        L.orc_sort.restype = ctypes.c_int
        L.orc_float_to_half.argtypes = [ctypes.c_double]
        L.orc_float_to_half.restype = ctypes.c_uint32
        L.orc_pack_half2x16.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_pack_half2x16.restype = ctypes.c_uint32
        L.orc_scene_pack.argtypes = [u8p, ctypes.c_uint32, u32p, f32p]
        L.orc_scene_pack.restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        L.orc_scene_build.argtypes = [u8p, ctypes.c_uint32, u32p, f32p, f32p, f32p]
        L.orc_scene_build.restype = None
        L.orc_scene_translate.argtypes = [ctypes.c_uint32, u32p, f32p, dp]
        L.orc_scene_translate.restype = None
        L.orc_scene_rotate.argtypes = [ctypes.c_uint32, u32p, f32p, f32p, f32p, dp]
        L.orc_scene_rotate.restype = None
        L.orc_scene_scale.argtypes = [ctypes.c_uint32, u32p, f32p, f32p, f32p, dp]
        L.orc_scene_scale.restype = None
        L.orc_scene_limit_box.argtypes = [ctypes.c_uint32, u32p, f32p, f32p, f32p, dp]
        L.orc_scene_limit_box.restype = ctypes.c_uint32
        L.orc_project.argtypes = [u32p, ctypes.c_uint32, f32p, f32p, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_int, ctypes.c_int, f32p, i32p, f32p]
        L.orc_project.restype = None
        L.orc_project_sh.argtypes = [u32p, ctypes.c_uint32, f32p, f32p, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_int, ctypes.c_int, u32p, u32p, u32p, i32p, f32p, i32p, f32p]
        L.orc_project_sh.restype = None
        L.orc_project_full.argtypes = [u32p, ctypes.c_uint32, f32p, f32p, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_int, ctypes.c_int, u32p, u32p, u32p, i32p, ctypes.c_int, ctypes.c_float,
                                       f32p, i32p, f32p]
        L.orc_project_full.restype = None
        L.orc_scene_pack_sh.argtypes = [f32p, ctypes.c_uint32, u32p, u32p, u32p]
        L.orc_scene_pack_sh.restype = None
        L.orc_tile_stats.argtypes = [i32p, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.orc_tile_stats.restype = None
        L.orc_render.argtypes = [ctypes.c_uint32, u32p, f32p, f32p, i32p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p]
        L.orc_render.restype = None
        L.orc_eval_sh_rgb.argtypes = [u32p, u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, f32p, f32p]
        L.orc_eval_sh_rgb.restype = None
        L.orc_fragment.argtypes = [f32p, f32p, f32p]
        L.orc_fragment.restype = ctypes.c_int
        L.orc_composite.argtypes = [f32p, ctypes.c_uint32, f32p]
        L.orc_composite.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def sort(vp, pos):
    """(depth_index u32[n], keys u32[n], (min,max)) -- wasm/wasm.cpp:8-52 restated."""
    vp = np.ascontiguousarray(vp, dtype=np.float32)
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1)
    n = pos.size // 3
    di = np.empty(n, dtype=np.uint32)
    keys = np.empty(n, dtype=np.uint32)
    mm = np.zeros(2, dtype=np.int32)
    rc = lib().orc_sort(_p(vp, f32p), n, _p(pos, f32p), _p(di, u32p), _p(keys, u32p), _p(mm, i32p))
    assert rc == 0
    return di, keys, (int(mm[0]), int(mm[1]))


def scene_pack(rows):
    """Scene.setData restated: (data u32[8n], positions f32[3n])."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1)
    n = rows.size // 32
    data = np.zeros(8 * n, dtype=np.uint32)
    pos = np.zeros(3 * n, dtype=np.float32)
    lib().orc_scene_pack(_p(rows, u8p), n, _p(data, u32p), _p(pos, f32p))
    return data, pos


class SceneState:
    """Scene.ts state (data, positions, rotations, scales) with the reference's transforms restated in f64."""

    def __init__(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1)
        self.n = rows.size // 32
        self.data = np.zeros(8 * self.n, dtype=np.uint32)
        self.positions = np.zeros(3 * self.n, dtype=np.float32)
        self.rotations = np.zeros(4 * self.n, dtype=np.float32)
        self.scales = np.zeros(3 * self.n, dtype=np.float32)
        lib().orc_scene_build(_p(rows, u8p), self.n, _p(self.data, u32p), _p(self.positions, f32p),
                              _p(self.rotations, f32p), _p(self.scales, f32p))

    @staticmethod
    def _d(v):
        return np.ascontiguousarray(v, dtype=np.float64)

    def translate(self, t):
        t = self._d(t)
        lib().orc_scene_translate(self.n, _p(self.data, u32p), _p(self.positions, f32p), t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))

    def rotate(self, q):
        q = self._d(q)
        lib().orc_scene_rotate(self.n, _p(self.data, u32p), _p(self.positions, f32p), _p(self.rotations, f32p), _p(self.scales, f32p),
                               q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))

    def scale(self, s):
        s = self._d(s)
        lib().orc_scene_scale(self.n, _p(self.data, u32p), _p(self.positions, f32p), _p(self.rotations, f32p), _p(self.scales, f32p),
                              s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))

    def limit_box(self, box):
        box = self._d(box)
        self.n = int(lib().orc_scene_limit_box(self.n, _p(self.data, u32p), _p(self.positions, f32p), _p(self.rotations, f32p),
                                               _p(self.scales, f32p), box.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        self.data, self.positions = self.data[:8 * self.n].copy(), self.positions[:3 * self.n].copy()
        self.rotations, self.scales = self.rotations[:4 * self.n].copy(), self.scales[:3 * self.n].copy()


def float_to_half(x):
    return int(lib().orc_float_to_half(float(x)))


def scene_pack_sh(shs):
    """Scene.setData's SH packing (Scene.ts:108-124): 48 floats per splat -> three u32[8*count] half textures."""
    shs = np.ascontiguousarray(shs, dtype=np.float32).reshape(-1)
    count = shs.size // 48
    out = [np.zeros(8 * count, dtype=np.uint32) for _ in range(3)]
    lib().orc_scene_pack_sh(_p(shs, f32p), count, _p(out[0], u32p), _p(out[1], u32p), _p(out[2], u32p))
    return out


def project(data, view, proj, fx, fy, W, H, sh=None, band=None, fade=None):
    """(rec f32[n,8] (col 7 holds rgb8 bits), bbox i32[n,4], raw f32[n,12]).  sh: three u32 arrays + band[3]."""
    data = np.ascontiguousarray(data, dtype=np.uint32).reshape(-1)
    n = data.size // 8
    view = np.ascontiguousarray(view, dtype=np.float32)
    proj = np.ascontiguousarray(proj, dtype=np.float32)
    rec = np.zeros((n, 8), dtype=np.float32)
    bbox = np.zeros((n, 4), dtype=np.int32)
    raw = np.zeros((n, 12), dtype=np.float32)
    if fade is not None:   # depth fade (FadeInPass): fade = u_depthFade
        null = ctypes.cast(None, u32p)
        shp = [null, null, null] if sh is None else [_p(np.ascontiguousarray(a, dtype=np.uint32), u32p) for a in sh]
        keep = None if sh is None else [np.ascontiguousarray(a, dtype=np.uint32) for a in sh]
        if keep is not None:
            shp = [_p(a, u32p) for a in keep]
        bandp = ctypes.cast(None, i32p) if band is None else _p(np.ascontiguousarray(band, dtype=np.int32), i32p)
        bkeep = None if band is None else np.ascontiguousarray(band, dtype=np.int32)
        if bkeep is not None:
            bandp = _p(bkeep, i32p)
        lib().orc_project_full(_p(data, u32p), n, _p(view, f32p), _p(proj, f32p), fx, fy, W, H, shp[0], shp[1], shp[2], bandp,
                               1, float(fade), _p(rec, f32p), _p(bbox, i32p), _p(raw, f32p))
    elif sh is None:
        lib().orc_project(_p(data, u32p), n, _p(view, f32p), _p(proj, f32p), fx, fy, W, H,
                          _p(rec, f32p), _p(bbox, i32p), _p(raw, f32p))
    else:
        sh = [np.ascontiguousarray(a, dtype=np.uint32) for a in sh]
        band = np.ascontiguousarray(band, dtype=np.int32)
        lib().orc_project_sh(_p(data, u32p), n, _p(view, f32p), _p(proj, f32p), fx, fy, W, H,
                             _p(sh[0], u32p), _p(sh[1], u32p), _p(sh[2], u32p), _p(band, i32p),
                             _p(rec, f32p), _p(bbox, i32p), _p(raw, f32p))
    return rec, bbox, raw


def eval_sh_rgb(sh, index, deg, direction):
    """eval_sh_rgb of the vertex shader (+ its min(rgb, 1)) for one splat of the three SH textures: f32[3]."""
    sh = [np.ascontiguousarray(a, dtype=np.uint32) for a in sh]
    d = np.ascontiguousarray(direction, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_eval_sh_rgb(_p(sh[0], u32p), _p(sh[1], u32p), _p(sh[2], u32p), int(index), int(deg), _p(d, f32p), _p(out, f32p))
    return out


def fragment(vpos, color):
    """The fragment shader for one fragment: premultiplied (B rgb, B) as f32[4], or None when it is discarded."""
    v = np.ascontiguousarray(vpos, dtype=np.float32)
    c = np.ascontiguousarray(color, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    return out if lib().orc_fragment(_p(v, f32p), _p(c, f32p), _p(out, f32p)) else None


def composite(frags):
    """k fragments (vPosition.xy, colour.rgba) blended in order onto one cleared pixel with the reference's blend state: f32[4]."""
    f = np.ascontiguousarray(frags, dtype=np.float32).reshape(-1, 6)
    out = np.zeros(4, dtype=np.float32)
    lib().orc_composite(_p(f, f32p), f.shape[0], _p(out, f32p))
    return out


def tile_stats(bbox, tile=16):
    bbox = np.ascontiguousarray(bbox, dtype=np.int32)
    V, D = ctypes.c_uint64(0), ctypes.c_uint64(0)
    lib().orc_tile_stats(_p(bbox, i32p), bbox.shape[0], tile, ctypes.byref(V), ctypes.byref(D))
    return V.value, D.value


def render(depth_index, raw, rec, bbox, W, H, mode=1, threads=None):
    """Premultiplied RGBA float32 image [H, W, 4], row 0 = top.  Row bands run on threads."""
    depth_index = np.ascontiguousarray(depth_index, dtype=np.uint32)
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    rec = np.ascontiguousarray(rec, dtype=np.float32)
    bbox = np.ascontiguousarray(bbox, dtype=np.int32)
    out = np.zeros((H, W, 4), dtype=np.float32)
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, H))
    edges = [H * t // threads for t in range(threads + 1)]
    L = lib()

    def band(t):
        L.orc_render(depth_index.size, _p(depth_index, u32p), _p(raw, f32p), _p(rec, f32p), _p(bbox, i32p),
                     W, H, mode, edges[t], edges[t + 1], _p(out, f32p))

    if threads == 1:
        band(0)
    else:
        ts = [threading.Thread(target=band, args=(t,)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    return out


def render_scene(data, pos, view, proj, viewproj, fx, fy, W, H, mode=1, threads=None, sh=None, band=None, fade=None):
    """Full oracle frame: (image, depth_index, V, D)."""
    di, _, _ = sort(viewproj, pos)
    rec, bbox, raw = project(data, view, proj, fx, fy, W, H, sh, band, fade)
    img = render(di, raw, rec, bbox, W, H, mode, threads)
    V, D = tile_stats(bbox)
    return img, di, V, D


# ---------------------------------------------------------------------------
# oracle/_ref: the reference's own wasm/wasm.cpp compiled from source
# ---------------------------------------------------------------------------
def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_sort.so"))


def ref_sort(vp, pos, calls=1):
    """Run the UNMODIFIED reference sort (wasm/wasm.cpp:8-52).

    The harness owns the scratch memory: `starts`/`counts` get 65537+ entries and
    starts[65536] is preset to N - #{q == 65536} before the call, which is the
    defined semantics for the reference's max-bucket overflow.
    Returns (depth_index, keys)."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(os.path.join(HERE, "_ref", "libref_sort.so"))
        _REF.sort.argtypes = [f32p, ctypes.c_uint32, f32p, u32p, u32p, u32p, u32p]
        _REF.sort.restype = None
    vp = np.ascontiguousarray(vp, dtype=np.float32)
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1)
    n = pos.size // 3
    m = max(n, 65536) + 8
    depth = np.zeros(m, dtype=np.uint32)
    di = np.zeros(m, dtype=np.uint32)
    starts = np.zeros(m, dtype=np.uint32)
    counts = np.zeros(m, dtype=np.uint32)

    def call():
        _REF.sort(_p(vp, f32p), n, _p(pos, f32p), _p(depth, u32p), _p(di, u32p), _p(starts, u32p), _p(counts, u32p))

    call()  # first call: learn how many splats land in the overflow bucket
    k = int(np.count_nonzero(depth[:n] == 65536))
    for _ in range(calls):
        starts[65536] = n - k
        di[:] = 0xFFFFFFFF
        call()
    return di[:n].copy(), depth[:n].copy()
