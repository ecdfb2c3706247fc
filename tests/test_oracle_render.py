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


# ---------------------------------------------------------------------------
# Off-axis, rotated, anisotropic splats against an INDEPENDENT float64 statement of the mathematics
# (tests/independent_math.py: linear algebra from what the shader's expressions mean, numpy eigensolver /
# matrix inverse; no shared operation sequence with oracle.c).  The centred isotropic KAT above cannot see a
# transposition or sign slip in J, transpose(mat3(view)) * J or Vrk (cam.x = cam.y = 0 zeroes J's third column and
# b = 0); these can: vertex.glsl.ts:146-175.
# ---------------------------------------------------------------------------
def _offaxis_case(oracle, rng, W, H, fx, fy):
    import independent_math as im
    eye = rng.normal(size=3)
    eye *= rng.uniform(5.0, 9.0) / np.linalg.norm(eye)
    target = rng.uniform(-0.5, 0.5, size=3)
    R = im.look_at_rotation(eye, target, rng.uniform(-np.pi, np.pi))
    view, proj, vp = im.camera_matrices(R, eye, fx, fy, W, H)
    p = target + rng.uniform(-1.2, 1.2, size=3)             # off the optical axis
    scale = np.exp(rng.uniform(np.log(0.02), np.log(0.35), size=3))
    scale[rng.integers(3)] *= 4.0                            # clearly anisotropic
    rot = tuple(int(v) for v in rng.integers(0, 256, size=4))
    rgba = tuple(int(v) for v in rng.integers(40, 256, size=4))
    data, pos = make_scene(oracle, [dict(pos=p, scale=scale, rgba=rgba, rot=rot)])
    p32 = pos[:3].astype(np.float64)
    centre, C, pc = im.footprint(p32, im.decode_cov4(data[:8]), view, fx, fy, W, H)
    return dict(view=view, proj=proj, vp=vp, data=data, pos=pos, centre=centre, C=C, pc=pc, rgba=rgba)


def _usable(case, W, H):
    a, b, c = case["C"][0, 0], case["C"][0, 1], case["C"][1, 1]
    cx, cy = case["centre"]
    return (case["pc"][2] > 1.0 and 0.15 * W < cx < 0.85 * W and 0.15 * H < cy < 0.85 * H and
            (a - c) ** 2 / 4 + b * b > 1.0 and abs(b) > 0.05 * (a + c) and 2 * max(a, c) < 900.0 ** 2 / 4)


def test_projection_matches_independent_f64_math(oracle):
    import independent_math as im
    W, H, fx, fy = 640, 480, 700.0, 650.0     # fx != fy on purpose
    rng = np.random.default_rng(2024)
    checked = 0
    for _ in range(200):
        case = _offaxis_case(oracle, rng, W, H, fx, fy)
        if not _usable(case, W, H):
            continue
        rec, bbox, raw = oracle.project(case["data"], case["view"], case["proj"], fx, fy, W, H)
        assert raw[0, 11] == 1.0
        major, minor, lam = im.axes_from_cov(case["C"])
        tol = 3e-4       # f32 chain of ~90 flops with cancellation in cov2d; a wrong sign or transposition is O(1)
        assert np.allclose(raw[0, 0:2], case["centre"], rtol=0, atol=2e-3), (raw[0, 0:2], case["centre"])
        assert np.linalg.norm(raw[0, 2:4] - major) <= tol * np.linalg.norm(major), (raw[0, 2:4], major)
        assert np.linalg.norm(raw[0, 4:6] - minor) <= tol * np.linalg.norm(major), (raw[0, 4:6], minor)
        # the derived record (image rows top-down): centre, and u / w with vPosition(p) = (dot(p-c,u), dot(p-c,w))
        flip = np.array([1.0, -1.0])
        u = 2 * (major * flip) / (major @ major)
        w = 2 * (minor * flip) / (minor @ minor)
        assert abs(rec[0, 0] - case["centre"][0]) < 2e-3 and abs(rec[0, 1] - (H - case["centre"][1])) < 2e-3
        assert np.linalg.norm(rec[0, 2:4] - u) <= tol * np.linalg.norm(u) * 4
        assert np.linalg.norm(rec[0, 4:6] - w) <= tol * np.linalg.norm(w) * 4
        assert abs(raw[0, 6] - case["rgba"][3] / 255.0) < 1e-7
        checked += 1
    assert checked >= 24, checked


@pytest.mark.parametrize("mode", [0, 1])
def test_offaxis_anisotropic_image_matches_closed_form(oracle, mode):
    import independent_math as im
    W, H, fx, fy = 160, 120, 260.0, 240.0
    rng = np.random.default_rng(77)
    checked = 0
    for _ in range(200):
        case = _offaxis_case(oracle, rng, W, H, fx, fy)
        a, c = case["C"][0, 0], case["C"][1, 1]
        if not _usable(case, W, H) or max(a, c) < 6.0:
            continue
        img, di, V, D = oracle.render_scene(case["data"], case["pos"], case["view"], case["proj"], case["vp"], fx, fy, W, H,
                                            mode=mode, threads=2)
        r, g, b, al = [v / 255.0 for v in case["rgba"]]
        want, edge = im.splat_image(case["centre"], case["C"], al, (r, g, b), W, H)
        err = np.abs(img.astype(np.float64) - want).max(axis=2)
        err[edge] = 0.0
        assert want[..., 3].max() > 0.3 * al and (want[..., 3] > 0).sum() > 40
        assert err.max() < 1e-4, (err.max(), case["C"])
        checked += 1
        if checked == 8:
            break
    assert checked == 8


def _overlap_case(oracle, rng, W, H, fx, fy, count):
    """`count` off-axis, rotated, anisotropic splats that overlap on screen, at well separated depths, under one
    camera -> (matrices, packed scene, per-splat closed forms) or None when a splat falls outside the usable range."""
    import independent_math as im
    eye = rng.normal(size=3)
    eye *= rng.uniform(5.0, 8.0) / np.linalg.norm(eye)
    target = rng.uniform(-0.4, 0.4, size=3)
    R = im.look_at_rotation(eye, target, rng.uniform(-np.pi, np.pi))
    view, proj, vp = im.camera_matrices(R, eye, fx, fy, W, H)
    fwd = R[:, 2]
    anchor = target + R[:, 0] * rng.uniform(-0.8, 0.8) + R[:, 1] * rng.uniform(-0.6, 0.6)   # off the optical axis
    depths = np.sort(rng.uniform(-1.5, 1.5, size=count))
    if np.diff(depths).min() < 0.12:
        return None
    splats = []
    for k in range(count):
        p = anchor + fwd * depths[k] + R[:, 0] * rng.uniform(-0.25, 0.25) + R[:, 1] * rng.uniform(-0.25, 0.25)
        scale = np.exp(rng.uniform(np.log(0.05), np.log(0.3), size=3))
        scale[rng.integers(3)] *= 3.0
        splats.append(dict(pos=p, scale=scale, rgba=tuple(int(v) for v in rng.integers(60, 256, size=4)),
                           rot=tuple(int(v) for v in rng.integers(0, 256, size=4))))
    order = rng.permutation(count)                       # scene order is not depth order: the sort has to do it
    data, pos = make_scene(oracle, [splats[i] for i in order])
    forms = []
    for j in range(count):
        centre, C, pc = im.footprint(pos[3 * j:3 * j + 3].astype(np.float64), im.decode_cov4(data[8 * j:8 * j + 8]), view, fx, fy, W, H)
        case = dict(centre=centre, C=C, pc=pc)
        if not _usable(case, W, H) or max(C[0, 0], C[1, 1]) < 6.0:
            return None
        rgba = [v / 255.0 for v in splats[order[j]]["rgba"]]
        forms.append(dict(centre=centre, C=C, opacity=rgba[3], rgb=rgba[:3], z=pc[2]))
    return dict(view=view, proj=proj, vp=vp, data=data, pos=pos, forms=forms)


def overlap_cases(oracle, want=6, W=160, H=120, fx=260.0, fy=240.0, seed=4242):
    """the fixed list of multi-splat cases both the oracle test (here) and the GPU test use"""
    import independent_math as im
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(4000):
        case = _overlap_case(oracle, rng, W, H, fx, fy, int(rng.integers(3, 6)))
        if case is None:
            continue
        img, edge = im.composite_under(case["forms"], W, H)
        # the splats must really overlap: somewhere at least three of them contribute
        layers = sum((im.splat_image(f["centre"], f["C"], f["opacity"], f["rgb"], W, H)[0][..., 3] > 0.02).astype(int) for f in case["forms"])
        if layers.max() < 3 or (layers >= 2).sum() < 60:
            continue
        case["want"], case["edge"] = img, edge
        out.append(case)
        if len(out) == want:
            break
    assert len(out) == want
    return out, (W, H, fx, fy)


@pytest.mark.parametrize("mode", [0, 1])
def test_overlapping_splats_composite_matches_independent_f64_under_blend(oracle, mode):
    """Multi-splat pin that does not come from oracle.c: 3-5 overlapping off-axis anisotropic splats per image, scene
    order shuffled, composited in float64 with the reference's "under" blend in ascending camera depth
    (independent_math.composite_under: WebGLRenderer.ts:282-285, frag.glsl.ts:13-21).  oracle modes 0 and 1 must agree
    within 1e-4 away from coverage edges; so must the HIP path (test_gpu_parity.py, same cases)."""
    cases, (W, H, fx, fy) = overlap_cases(oracle)
    for case in cases:
        img, di, V, D = oracle.render_scene(case["data"], case["pos"], case["view"], case["proj"], case["vp"], fx, fy, W, H,
                                            mode=mode, threads=2)
        assert V == len(case["forms"])
        # the sort's order is the stated one: ascending camera depth
        assert list(di) == list(np.argsort([f["z"] for f in case["forms"]], kind="stable"))
        err = np.abs(img.astype(np.float64) - case["want"]).max(axis=2)
        err[case["edge"]] = 0.0
        assert err.max() < 1e-4, err.max()
        # and the order matters in these cases: back to front would be a different image
        back = np.zeros_like(case["want"])
        import independent_math as im
        for f in sorted(case["forms"], key=lambda s: -s["z"]):
            src, _ = im.splat_image(f["centre"], f["C"], f["opacity"], f["rgb"], W, H)
            back += (1.0 - back[..., 3:4]) * src
        assert np.abs(back - case["want"]).max() > 1e-2


def test_rgba8_rop_mode_quantifies_the_canvas_gap(oracle, scenes):
    """oracle mode 2 re-quantises the destination to RGBA8 after every fragment, like the reference's default drawing
    buffer (WebGLRenderer.ts:38,139-142,282-285).  It is a model used to QUANTIFY how far a browser canvas is from the
    fp32 image parity is defined on (scripts/rop_gap_report.py), not a parity target."""
    from gsplat_hip import orbit_camera, synth
    cfg = synth.CONFIGS["C1"]
    rows, data, pos = scenes("C1")
    cam = orbit_camera(40, width=cfg["width"], height=cfg["height"], fx=cfg["fx"])
    a = render(oracle, data, pos, cam, cfg["width"], cfg["height"], 0)[0].astype(np.float64)
    b = render(oracle, data, pos, cam, cfg["width"], cfg["height"], 2)[0].astype(np.float64)
    assert np.abs(b * 255 - np.round(b * 255)).max() < 1e-4          # every value is an 8-bit level
    d = np.abs(a - b) * 255
    assert 0.2 < d.mean() < 1.0 and 2.0 < d.max() < 12.0             # C1: mean ~0.5 LSB, max ~4-5 LSB
    # a single fragment is quantised exactly once: within half an LSB of the fp32 value
    d1, p1 = make_scene(oracle, [dict(pos=(0.2, 0.1, 0), scale=(0.2, 0.05, 0.1), rgba=(200, 100, 50, 180), rot=(200, 160, 90, 130))])
    c1 = front_camera(96, 96, 300.0)
    x = render(oracle, d1, p1, c1, 96, 96, 0)[0].astype(np.float64)
    y = render(oracle, d1, p1, c1, 96, 96, 2)[0].astype(np.float64)
    assert np.abs(x - y).max() <= 0.5 / 255 + 1e-7 and y.max() > 0.3
