"""numpy restatement of the reference's PLY -> .splat conversion (src/loaders/PLYLoader.ts), used only by
tests/test_js_host.py to check the JavaScript PLYLoader.  TEST INFRASTRUCTURE ONLY.

  rows_from_ply      _ParsePLYBuffer            PLYLoader.ts:389-538  (format "" or "polycam")
  rows_and_sh_from_ply  _ParseFullPLYBufferFast PLYLoader.ts:578-712  (with its two oddities, see gsplat.js_amd/js/loaders/PLYLoader.js)
  rows_sh_from_qply  _ParseQPLYBuffer           PLYLoader.ts:893-1196 (codebook-quantized variant; half decode utils.ts:52-71)
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


# ---------------------------------------------------------------------------
# quantized PLY (PLYLoader.ts:893-1196)
# ---------------------------------------------------------------------------
QPLY_CODEBOOKS = ["features_dc"] + ["features_rest_%d" % i for i in range(15)] + ["opacity", "scaling", "rotation_re", "rotation_im"]
_SIZES = dict(double=8, int=4, uint=4, float=4, short=2, ushort=2, uchar=1)


def _half(bits):
    """utils.ts:52-71 on uint16 bit patterns -> float64 (the reference stores float32: exact)."""
    return np.asarray(bits, dtype=np.uint16).view(np.float16).astype(np.float64)


def rows_sh_from_qply(buf):
    import re
    text = bytes(buf[:10240]).decode("utf8", errors="replace")
    end = text.index("end_header\n")
    body = end + len("end_header\n")
    cb_start = text.index("element codebook_centers 256\n")
    ms = list(re.finditer(r"element vertex_(\d+) (\d+)", text))
    counts = [int(m.group(2)) for m in ms]
    starts = [m.start() for m in ms]
    extents = [(0, starts[1]), (starts[1], starts[2]), (starts[2], starts[3]), (starts[3], cb_start)]
    props, row = [], []
    for a, b in extents:
        d, off = {}, 0
        for line in text[a:b].split("\n"):
            if line.startswith("property "):
                _, typ, name = line.split(" ")
                d[name] = off
                off += _SIZES[typ]
        props.append(d)
        row.append(off)
    data_bytes = sum(c * r for c, r in zip(counts, row))
    names = [line.split(" ")[2] for line in text[cb_start:end].split("\n") if line.startswith("property ")]
    nb = len(names)
    raw = np.frombuffer(buf, dtype="<u2", count=256 * nb, offset=body + data_bytes).reshape(256, nb)
    cb = {name: _half(raw[:, j]).astype(np.float32).astype(np.float64) for j, name in enumerate(names)}
    total = sum(counts)
    rows = np.zeros((total, 32), dtype=np.uint8)
    sh = np.zeros((counts[1] + counts[2] + counts[3], 48), dtype=np.float32)
    rest0 = props[1]["f_rest_0"]
    w = r = s_off = 0
    for e in range(4):
        n, pr, rs = counts[e], props[e], row[e]
        blk = np.frombuffer(buf, dtype=np.uint8, count=n * rs, offset=body + r).reshape(n, rs) if n else np.zeros((0, max(rs, 1)), np.uint8)
        u8 = lambda name: blk[:, pr[name]]
        half_at = lambda name: _half(blk[:, pr[name]].astype(np.uint16) | (blk[:, pr[name] + 1].astype(np.uint16) << 8))
        out = rows[w:w + n]
        pos = np.stack([half_at("x"), half_at("y"), half_at("z")], axis=1).astype(np.float32)
        out[:, 0:12] = pos.view(np.uint8).reshape(n, 12)
        scale = np.exp(np.stack([cb["scaling"][u8("scale_%d" % k)] for k in range(3)], axis=1)).astype(np.float32)
        out[:, 12:24] = scale.view(np.uint8).reshape(n, 12)
        for c in range(3):
            out[:, 24 + c] = _clamp8((0.5 + SH_C0 * cb["features_dc"][u8("f_dc_%d" % c)]) * 255)
        out[:, 27] = _clamp8((1 / (1 + np.exp(-cb["opacity"][u8("opacity")]))) * 255)
        if n:
            out[:, 28:32] = _rot_bytes(cb["rotation_re"][u8("rot_0")], cb["rotation_im"][u8("rot_1")], cb["rotation_im"][u8("rot_2")],
                                       cb["rotation_im"][u8("rot_3")])
        if e > 0:
            stride = [3, 8, 15][e - 1]
            nrest = sum(1 for k in pr if k.startswith("f_rest"))
            o = sh[s_off:s_off + n]
            for c in range(3):
                o[:, c] = cb["features_dc"][u8("f_dc_%d" % c)]
            for m in range(nrest):
                coef = m // 3
                o[:, 3 + m] = cb["features_rest_%d" % coef][blk[:, rest0 + coef + stride * (m % 3)]]
            s_off += n
        w += n
        r += n * rs
    ind0 = counts[0] - 1
    return rows.reshape(-1), sh.reshape(-1), np.array([ind0, ind0 + counts[1], ind0 + counts[1] + counts[2]], dtype=np.int32)


def synth_qply(counts, seed):
    """A quantized PLY with the layout _ParseQPLYBuffer expects: vertex_0..3 (0..3 SH bands), half positions, uchar
    codebook indices, then 256 x 20 half codebook entries."""
    rng = np.random.default_rng(seed)
    header = "ply\nformat binary_little_endian 1.0\n"
    blobs = []
    for e, n in enumerate(counts):
        nrest = [0, 9, 24, 45][e]
        header += "element vertex_%d %d\n" % (e, n)
        header += "".join("property short %s\n" % a for a in "xyz")
        names = ["f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(nrest)] + ["opacity", "scale_0", "scale_1", "scale_2",
                                                                                         "rot_0", "rot_1", "rot_2", "rot_3"]
        header += "".join("property uchar %s\n" % a for a in names)
        pos = (rng.standard_normal((n, 3)) * 2).astype(np.float16).view(np.uint16).astype("<u2")
        idx = rng.integers(0, 256, (n, len(names)), dtype=np.uint8)
        rowb = np.zeros((n, 6 + len(names)), dtype=np.uint8)
        rowb[:, :6] = pos.view(np.uint8).reshape(n, 6)
        rowb[:, 6:] = idx
        blobs.append(rowb.tobytes())
    header += "element codebook_centers 256\n" + "".join("property short %s\n" % c for c in QPLY_CODEBOOKS) + "end_header\n"
    cbv = rng.standard_normal((256, len(QPLY_CODEBOOKS)))
    cbv[:, QPLY_CODEBOOKS.index("scaling")] = rng.uniform(-6, -2, 256)
    cbv[:, QPLY_CODEBOOKS.index("opacity")] = rng.uniform(-4, 6, 256)
    cbv[0, 0] = 0.0; cbv[1, 0] = -0.0; cbv[2, 0] = 6e-8; cbv[3, 0] = 65504.0     # zero, minus zero, a subnormal, the largest half
    cbb = cbv.astype(np.float16).view(np.uint16).astype("<u2").tobytes()
    return header.encode() + b"".join(blobs) + cbb
