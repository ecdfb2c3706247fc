"""numpy restatement of the reference's PLY -> .splat conversion (src/loaders/PLYLoader.ts), used only by
tests/test_js_host.py to check the JavaScript PLYLoader.  TEST INFRASTRUCTURE ONLY.

  rows_from_ply      _ParsePLYBuffer            PLYLoader.ts:389-538  (format "" or "polycam")
  rows_and_sh_from_ply  _ParseFullPLYBufferFast PLYLoader.ts:578-712  (with its two oddities, see gsplat.js_amd/js/loaders/PLYLoader.js)
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def parse_header(buf):
    text = bytes(buf[:10240]).decode("utf8", errors="replace")
    end = text.index("end_header\n")
    n = int(text.split("element vertex ")[1].split("\n")[0])
    props, off = [], 0
    sizes = dict(double=8, int=4, uint=4, float=4, short=2, ushort=2, uchar=1)
    for line in text[:end].split("\n"):
        if line.startswith("property "):
            _, typ, name = line.split(" ")
            props.append((name, typ, off))
            off += sizes[typ]
    return props, end + len("end_header\n"), n, off


def _clamp8(x):
    """ToUint8Clamp: NaN -> 0, clamp, round half to even."""
    x = np.where(np.isnan(x), 0.0, x)
    return np.rint(np.clip(x, 0.0, 255.0)).astype(np.uint8)


def _column(buf, start, n, stride, off, typ):
    dt = {"float": "<f4", "int": "<i4"}[typ]
    return np.ndarray((n,), dtype=dt, buffer=buf, offset=start + off, strides=(stride,)).astype(np.float64)


def _rot_bytes(r0, r1, r2, r3, pre=None):
    x, y, z, w = r1, r2, r3, r0
    if pre is not None:   # q_polycam.multiply(q)  (Quaternion.ts:39-55)
        x1, y1, z1, w1 = pre
        x, y, z, w = (w1 * x + x1 * w + y1 * z - z1 * y, w1 * y - x1 * z + y1 * w + z1 * x,
                      w1 * z + x1 * y - y1 * x + z1 * w, w1 * w - x1 * x - y1 * y - z1 * z)
    l = np.sqrt(x * x + y * y + z * z + w * w)
    return np.stack([_clamp8(w / l * 128 + 128), _clamp8(x / l * 128 + 128), _clamp8(y / l * 128 + 128), _clamp8(z / l * 128 + 128)], axis=1)


def rows_from_ply(buf, fmt=""):
    props, start, n, stride = parse_header(buf)
    col = {name: _column(buf, start, n, stride, off, typ) for name, typ, off in props}
    rows = np.zeros((n, 32), dtype=np.uint8)
    pos = np.stack([col["x"], col["y"], col["z"]], axis=1).astype(np.float32)
    pre = None
    if fmt == "polycam":
        pos = np.stack([pos[:, 0], -pos[:, 2], pos[:, 1]], axis=1)
        h = math.pi / 2 / 2   # Quaternion.FromEuler(Vector3(pi/2, 0, 0))
        pre = (math.cos(0) * math.sin(h) * math.cos(0) + 0.0, 0.0 - math.cos(0) * math.sin(h) * 0.0, 0.0, math.cos(h))
        pre = (math.sin(h), 0.0, 0.0, math.cos(h))
    rows[:, 0:12] = pos.view(np.uint8).reshape(n, 12)
    scale = np.exp(np.stack([col["scale_0"], col["scale_1"], col["scale_2"]], axis=1)).astype(np.float32)
    rows[:, 12:24] = scale.view(np.uint8).reshape(n, 12)
    for c in range(3):
        rows[:, 24 + c] = _clamp8((0.5 + SH_C0 * col["f_dc_%d" % c]) * 255)
    rows[:, 27] = _clamp8((1 / (1 + np.exp(-col["opacity"]))) * 255)
    rows[:, 28:32] = _rot_bytes(col["rot_0"], col["rot_1"], col["rot_2"], col["rot_3"], pre)
    return rows.reshape(-1)


def rows_and_sh_from_ply(buf):
    props, start, n, _ = parse_header(buf)
    stride = props[-1][2] + 4
    col = {name: _column(buf, start, n, stride, off, "float") for name, typ, off in props}
    rows = np.zeros((n, 32), dtype=np.uint8)
    rows[:, 0:12] = np.stack([col["x"], col["y"], col["z"]], axis=1).astype(np.float32).view(np.uint8).reshape(n, 12)
    scale = np.exp(np.stack([col["scale_0"], col["scale_1"], col["scale_2"]], axis=1)).astype(np.float32)
    rows[:, 12:24] = scale.view(np.uint8).reshape(n, 12)
    for c in range(3):
        rows[:, 24 + c] = _clamp8(0.5 + SH_C0 * col["f_dc_%d" % c] * 255)    # precedence as written at :617-619
    rows[:, 27] = _clamp8((1 / (1 + np.exp(-col["opacity"]))) * 255)
    rows[:, 28:32] = _rot_bytes(col["rot_0"], col["rot_1"], col["rot_2"], col["rot_3"])
    order = [k + 15 * c for k in range(15) for c in range(3)]
    order[29] = 38                                                            # :689 repeats f_rest_38 in the slot of f_rest_39
    sh = np.zeros((n, 48), dtype=np.float32)
    for c in range(3):
        sh[:, c] = col["f_dc_%d" % c]
    for j, src in enumerate(order):
        sh[:, 3 + j] = col["f_rest_%d" % src]
    return rows.reshape(-1), sh.reshape(-1)


def synth_ply(n, seed):
    """A binary little-endian PLY with the INRIA 3DGS property list and random contents."""
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] + \
            ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]
    rng = np.random.default_rng(seed)
    body = rng.standard_normal((n, len(names))).astype(np.float32)
    body[:, names.index("scale_0"):names.index("scale_0") + 3] = rng.uniform(-6, -2, (n, 3))
    body[:, names.index("opacity")] = rng.uniform(-4, 6, n)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % s for s in names) + "end_header\n"
    return header.encode() + body.tobytes()
