"""Known-answer tests of the render oracle (G6 of SURVEY.md 8(c)).  The reference's GLSL cannot run
here, so these pin the restatement to closed forms derived from vertex.glsl.ts / frag.glsl.ts."""
import math

import numpy as np
import pytest


def make_scene(oracle, splats):
    """splats: list of dict(pos, scale, rgba, rot=(w,x,y,z) bytes)."""
    rows = np.zeros((len(splats), 32), dtype=np.uint8)
    for i, s in enumerate(splats):
        rows[i, 0:12] = np.asarray(s["pos"], dtype=np.float32).view(np.uint8)
        rows[i, 12:24] = np.asarray(s["scale"], dtype=np.float32).view(np.uint8)
        rows[i, 24:28] = s["rgba"]
        rows[i, 28:32] = s.get("rot", (255, 128, 128, 128))
    return oracle.scene_pack(rows.reshape(-1))


def front_camera(W, H, fx=500.0, z=-5.0):
    from gsplat_hip import Camera
    return Camera((0.0, 0.0, z), (0.0, 0.0, 0.0, 1.0), fx, fx).update(W, H)


def render(oracle, data, pos, cam, W, H, mode):
    v, p, vp = cam.f32()
    return oracle.render_scene(data, pos, v, p, vp, cam.fx, cam.fy, W, H, mode=mode, threads=2)


@pytest.mark.parametrize("mode", [0, 1])
def test_single_isotropic_splat_closed_form(oracle, mode):
    W = H = 129
    s, fx, z = 0.05, 500.0, 5.0
    data, pos = make_scene(oracle, [dict(pos=(0, 0, 0), scale=(s, s, s), rgba=(255, 128, 0, 255))])
    img, di, V, D = render(oracle, data, pos, front_camera(W, H, fx, -z), W, H, mode)
    # 4*Sigma = 4 s^2 I (half-truncated); cov2d = ((fx/z)^2 * 4 s^2 + 0.3) I =: lam I, b = 0.
    # vertex.glsl.ts:166-168: sqrt(max(0.1, mid^2 - det)) = sqrt(0.1) even for a circular footprint, so
    # lambda1,2 = lam +- sqrt(0.1); diagonalVector = normalize(0, lambda1 - a) = (0, 1): major axis vertical.
    # vPosition = (2 dy / sqrt(2 lambda1), 2 dx / sqrt(2 lambda2)); A = -2 (dy^2/lambda1 + dx^2/lambda2)
    from gsplat_hip import _float_to_half
    h = int(_float_to_half(np.array([4 * np.float32(s) * np.float32(s)]))[0])
    sig4 = (1 + (h & 0x3FF) / 1024.0) * 2.0 ** ((h >> 10) - 15)
    lam = (fx / z) ** 2 * sig4 + 0.3
    l1, l2 = lam + math.sqrt(0.1), lam - math.sqrt(0.1)
    cx = cy = W / 2.0
    ys, xs = np.mgrid[0:H, 0:W]
    A = -2.0 * ((ys + 0.5 - cy) ** 2 / l1 + (xs + 0.5 - cx) ** 2 / l2)
    B = np.where(A < -4.0, 0.0, np.exp(A) * 1.0)
    edge = np.abs(A + 4.0) < 1e-3      # pixel centres within rounding of the ellipse edge may flip
    img = img.copy(); img[edge] = 0; B[edge] = 0
    assert V == 1
    assert np.abs(img[..., 3] - B).max() < 2e-5
    assert np.abs(img[..., 0] - B).max() < 2e-5 and np.abs(img[..., 1] - B * (128 / 255)).max() < 2e-5
    assert img[..., 2].max() == 0 and B.max() > 0.9


@pytest.mark.parametrize("mode", [0, 1])
def test_two_overlapping_splats_front_to_back(oracle, mode):
    W = H = 65
    near = dict(pos=(0, 0, -1.0), scale=(0.05,) * 3, rgba=(255, 0, 0, 128))   # closer to the camera at z=-5
    far = dict(pos=(0, 0, 1.0), scale=(0.05,) * 3, rgba=(0, 255, 0, 255))
    for order in ([near, far], [far, near]):   # storage order must not matter: depth sort decides
        data, pos = make_scene(oracle, order)
        img, di, V, D = render(oracle, data, pos, front_camera(W, H), W, H, mode)
        c = img[H // 2, W // 2]
        # dst = near + (1 - a_near) * far ; pixel centre is 0.5 px off the splat centre -> weights just below alpha
        a_n, a_f = c[0], c[1] / (1 - c[0])
        assert 0.49 < a_n < 128 / 255 + 1e-6
        assert 0.97 < a_f <= 1.0 + 1e-6
        assert abs(c[3] - (a_n + (1 - a_n) * a_f)) < 1e-6
        assert order[di[0]] is near


def test_culls_and_drops(oracle):
    W, H, fx = 200, 100, 100.0
    cam = front_camera(W, H, fx, -5.0)
    v, p, vp = cam.f32()
    sc = (0.05,) * 3
    # x_ndc = 2 fx/W * X/Z: at Z=5 X=5.5 -> 1.1 (inside 1.2x), X=6.5 -> 1.3 (culled); behind camera culled
    data, pos = make_scene(oracle, [
        dict(pos=(5.5, 0.7, 0), scale=sc, rgba=(255, 255, 255, 255)),
        dict(pos=(6.5, 0.7, 0), scale=sc, rgba=(255, 255, 255, 255)),
        dict(pos=(0, 0, -10.0), scale=sc, rgba=(255, 255, 255, 255)),
        dict(pos=(0, 0, 0), scale=(0.0, 0.0, 0.0), rgba=(255, 255, 255, 255)),   # cov2d = 0.3 I: lambda2 = 0.3 - sqrt(0.1) < 0 -> dropped (:171)
        dict(pos=(0.3, 0.2, 0), scale=(0.5, 0.01, 0.01), rgba=(255, 255, 255, 255), rot=(200, 160, 90, 130)),
    ])
    rec, bbox, raw = oracle.project(data, v, p, fx, fx, W, H)
    assert raw[0, 11] == 1 and bbox[0, 0] > bbox[0, 2]          # survives the 1.2x cull, but off-screen
    assert raw[1, 11] == 0 and raw[2, 11] == 0
    assert raw[3, 11] == 0                                       # lambda2 < 0 -> dropped
    assert raw[4, 11] == 1 and bbox[4, 0] <= bbox[4, 2]
    # reference quirk: b == 0 and lambda1 == a  ->  normalize(vec2(0,0)) = NaN  ->  splat dropped.
    # An isotropic splat displaced along x only has cov2d = diag(a > c), b = 0: the reference loses it.
    d2, p2 = make_scene(oracle, [dict(pos=(4.0, 0, 0), scale=sc, rgba=(255, 255, 255, 255))])
    assert oracle.project(d2, v, p, fx, fx, W, H)[2][0, 11] == 0
    maj, mnr = raw[4, 2:4], raw[4, 4:6]
    assert abs(float(maj @ mnr)) < 1e-3 * np.linalg.norm(maj) * np.linalg.norm(mnr)   # axes orthogonal


def test_axis_clamp_1024(oracle):
    W, H, fx = 256, 256, 2000.0
    cam = front_camera(W, H, fx, -1.0)
    v, p, vp = cam.f32()
    data, pos = make_scene(oracle, [dict(pos=(0.05, 0.02, 0), scale=(2.0, 1.5, 1.0), rgba=(10, 20, 30, 40), rot=(200, 160, 90, 130))])
    rec, bbox, raw = oracle.project(data, v, p, fx, fx, W, H)
    assert raw[0, 11] == 1
    assert abs(np.linalg.norm(raw[0, 2:4]) - 1024.0) < 1e-2 and abs(np.linalg.norm(raw[0, 4:6]) - 1024.0) < 1e-2
    assert bbox[0].tolist() == [0, 0, W - 1, H - 1]


def test_restated_and_ideal_modes_agree_except_boundary_flips(oracle, scenes):
    from gsplat_hip import orbit_camera, synth
    cfg = synth.CONFIGS["C1"]
    rows, data, pos = scenes("C1")
    cam = orbit_camera(11, width=cfg["width"], height=cfg["height"], fx=cfg["fx"])
    a = render(oracle, data, pos, cam, cfg["width"], cfg["height"], 0)[0]
    b = render(oracle, data, pos, cam, cfg["width"], cfg["height"], 1)[0]
    d = np.abs(a - b).max(axis=2)
    # identical up to f32 rounding, except pixels whose centre lies within rounding of an ellipse edge
    assert (d > 2e-4).sum() <= 16
    assert d.max() < math.exp(-4.0) + 1e-3


def test_tile_stats(oracle):
    bbox = np.array([[0, 0, 15, 15], [15, 15, 16, 16], [1, 1, 0, 0], [0, 0, 639, 0]], dtype=np.int32)
    assert oracle.tile_stats(bbox, 16) == (3, 1 + 4 + 40)
