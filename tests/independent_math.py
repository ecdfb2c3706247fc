"""Independent float64 statement of the projection + fragment MATHEMATICS of the reference's splat pass, used to pin
`oracle/oracle.c` (and through it the HIP kernels) on off-axis, rotated, anisotropic splats.

It is deliberately NOT a transcription of the shader's operation sequence (that is what oracle.c is): everything is
written as linear algebra with numpy in the textbook orientation, derived from what the shader's expressions mean
(/root/reference/src/renderers/webgl/shaders/vertex.glsl.ts:133-175,226-229, frag.glsl.ts:13-21, Camera.ts:32-56,81-92):

  p_cam  = R^T (p - t)                        camera looks down +z, R = camera-to-world rotation, t = camera position
  ndc    = (2 fx / W * x/z, -2 fy / H * y/z)  -> window (GL, y up): (W/2 + fx x/z, H/2 - fy y/z)
  J'     = d(window)/d(p_cam) = [[fx/z, 0, -fx x/z^2], [0, -fy/z, fy y/z^2]]
  C      = (J' Wv) (4 Sigma) (J' Wv)^T + 0.3 I          Wv = R^T (rotation part of the view matrix)
           GLSL: mat3(...) lists COLUMNS, so the shader's `J` is J'^T and `T = transpose(mat3(view)) * J` = (J' Wv)^T
  a fragment at window offset d from the centre has vPosition v with d = (v.x * major + v.y * minor) / 2,
  major/minor = sqrt(2 lambda_k) e_k (eigenpairs of C)  =>  |v|^2 = 2 d^T C^-1 d
  weight B = opacity * exp(-2 d^T C^-1 d) where 2 d^T C^-1 d <= 4, nothing outside.

The shader replaces sqrt(mid^2 - det) by sqrt(max(0.1, .)), which differs from the true eigenvalues only for nearly
circular footprints; callers keep to splats with (a-c)^2/4 + b^2 > 0.1 so that the closed form is exact.
"""
import numpy as np


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def look_at_rotation(eye, target, roll, rng=None):
    """camera-to-world rotation whose +z axis points from eye to target, rolled about it by `roll` radians"""
    f = np.asarray(target, float) - np.asarray(eye, float)
    f /= np.linalg.norm(f)
    up = np.array([0.0, 1.0, 0.0]) if abs(f[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    r = np.cross(up, f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    c, s = np.cos(roll), np.sin(roll)
    r2, u2 = c * r + s * u, -s * r + c * u
    return np.stack([r2, u2, f], axis=1)   # columns = camera axes in world coordinates


def camera_matrices(R, t, fx, fy, W, H, near=0.01, far=1000.0):
    """column-major float32[16] view / projection / viewProj as the renderer uploads them, from first principles:
    view = [R^T | -R^T t]; projection maps (x, y, z, 1)_cam to clip (2fx/W x, -2fy/H y, far/(far-near) z - far near/(far-near), z)."""
    V = np.eye(4)
    V[:3, :3] = R.T
    V[:3, 3] = -R.T @ np.asarray(t, float)
    P = np.zeros((4, 4))
    P[0, 0] = 2 * fx / W
    P[1, 1] = -2 * fy / H
    P[2, 2] = far / (far - near)
    P[2, 3] = -far * near / (far - near)
    P[3, 2] = 1.0
    VP = P @ V
    cm = lambda M: np.ascontiguousarray(M.T.reshape(-1), dtype=np.float32)   # column-major flattening
    return cm(V), cm(P), cm(VP)


def decode_cov4(data_row):
    """4*Sigma (3x3 symmetric, float64) from words 4..6 of a Scene.data row: six IEEE halves."""
    h = np.array([data_row[4] & 0xFFFF, data_row[4] >> 16, data_row[5] & 0xFFFF, data_row[5] >> 16,
                  data_row[6] & 0xFFFF, data_row[6] >> 16], dtype=np.uint16).view(np.float16).astype(np.float64)
    return np.array([[h[0], h[1], h[2]], [h[1], h[3], h[4]], [h[2], h[4], h[5]]])


def footprint(p, cov4, view32, fx, fy, W, H):
    """(centre_window_xy (GL, y up), C 2x2, p_cam) for one splat.  view32: the float32 view matrix the renderer gets
    (the shader works on those rounded values, so they are the inputs here too; all arithmetic after that is f64)."""
    V = view32.astype(np.float64).reshape(4, 4).T
    pc = V[:3, :3] @ np.asarray(p, np.float64) + V[:3, 3]
    x, y, z = pc
    centre = np.array([W / 2.0 + fx * x / z, H / 2.0 - fy * y / z])
    Jp = np.array([[fx / z, 0.0, -fx * x / (z * z)], [0.0, -fy / z, fy * y / (z * z)]])
    A = Jp @ V[:3, :3]
    C = A @ cov4 @ A.T + 0.3 * np.eye(2)
    return centre, C, pc


def axes_from_cov(C):
    """(major, minor) as the shader defines them: sqrt(2 lambda) * eigenvector, major's y component >= 0,
    minor = sqrt(2 lambda2) * (e.y, -e.x); via numpy's symmetric eigensolver, not the shader's closed form."""
    lam, vec = np.linalg.eigh(C)
    e = vec[:, 1]
    if e[1] < 0:
        e = -e
    return np.sqrt(2 * lam[1]) * e, np.sqrt(2 * lam[0]) * np.array([e[1], -e[0]]), lam[::-1]


def splat_image(centre, C, opacity, rgb, W, H):
    """premultiplied RGBA float64 [H, W, 4] of ONE splat on a clear canvas, row 0 = top, plus the mask of pixels
    whose centre lies within 1e-3 of the coverage edge (where f32 evaluations may legitimately flip)."""
    ys, xs = np.mgrid[0:H, 0:W]
    dx = (xs + 0.5) - centre[0]
    dy = (H - (ys + 0.5)) - centre[1]      # GL window rows run bottom-up
    Ci = np.linalg.inv(C)
    q = 2.0 * (Ci[0, 0] * dx * dx + 2 * Ci[0, 1] * dx * dy + Ci[1, 1] * dy * dy)
    B = np.where(q <= 4.0, opacity * np.exp(-q), 0.0)
    img = np.zeros((H, W, 4))
    for k in range(3):
        img[..., k] = B * rgb[k]
    img[..., 3] = B
    return img, np.abs(q - 4.0) < 1e-3


def composite_under(splats, W, H):
    """premultiplied RGBA float64 [H, W, 4] of SEVERAL splats composited front to back with the reference's blend state
    (WebGLRenderer.ts:139-142,282-285: blendFuncSeparate(ONE_MINUS_DST_ALPHA, ONE, ONE_MINUS_DST_ALPHA, ONE), FUNC_ADD,
    cleared to 0), i.e. per fragment in draw order   dst.rgb += (1 - dst.a) * src.rgb;  dst.a += (1 - dst.a) * src.a
    with src = (B * rgb, B) from frag.glsl.ts:13-21.  `splats`: dicts with centre, C, opacity, rgb and z (camera-space
    depth); the draw order is the sort's front-to-back order, stated here as ascending z (wasm.cpp:14-51 sorts by row 2 of
    viewProj, which is z times far/(far-near): callers keep the depths well apart, so the 16-bit quantisation cannot
    tie them).  Also returns the union of the per-splat coverage-edge masks."""
    img = np.zeros((H, W, 4))
    edge = np.zeros((H, W), dtype=bool)
    for sp in sorted(splats, key=lambda s: s["z"]):
        src, e = splat_image(sp["centre"], sp["C"], sp["opacity"], sp["rgb"], W, H)
        img += (1.0 - img[..., 3:4]) * src
        edge |= e
    return img, edge
