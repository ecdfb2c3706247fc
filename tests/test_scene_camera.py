"""Host logic: Scene.setData packing and Camera.update matrices (Python harness mirror),
against the oracle's C restatement and the fixtures produced by running the reference's
floatToHalf under Node (tests/golden/make_golden_half.js)."""
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_float_to_half_matches_reference_fixture(oracle):
    import gsplat_hip as gh
    g = json.load(open(os.path.join(HERE, "golden", "half_golden.json")))
    xs = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in g["inputs_f64_hex"]])
    want = np.array(g["half"], dtype=np.int64) & 0xFFFFFFFF
    got_oracle = np.array([oracle.float_to_half(x) for x in xs], dtype=np.int64)
    assert np.array_equal(got_oracle, want)
    assert np.array_equal(gh._float_to_half(xs).astype(np.int64), want)
    pairs = np.array(g["pack_pairs"], dtype=np.int64) & 0xFFFFFFFF
    m = pairs.size
    assert np.array_equal(gh.pack_half2x16(xs[0:2 * m:2], xs[1:2 * m:2]).astype(np.int64), pairs)


def test_float_to_half_documented_quirks(oracle):
    # truncation (no rounding), >= 32768 -> inf, JS shift-count-mod-32 garbage below 2^-46
    assert oracle.float_to_half(1.0) == 0x3C00
    assert oracle.float_to_half(0.3) == 0x34CC            # truncated (RNE would give 0x34CD)
    assert oracle.float_to_half(32768.0) == 0x7C00
    assert oracle.float_to_half(65504.0) == 0x7C00        # the largest finite half is unreachable
    assert oracle.float_to_half(float("nan")) == 0x7C00
    assert oracle.float_to_half(-0.0) == 0x8000
    assert oracle.float_to_half(1e-15) == 0x0048
    assert oracle.float_to_half(1e-38) == 0x0367


def test_scene_setdata_matches_oracle(oracle):
    import gsplat_hip as gh
    rows = gh.synth.synth_rows(20000, 77)
    # force a few edge rows: identity rotation, zero scale, huge scale, alpha 0
    r = rows.reshape(-1, 32).copy()
    r[0, 28:32] = [255, 128, 128, 128]
    r[1, 12:24] = np.zeros(3, dtype=np.float32).view(np.uint8)
    r[2, 12:24] = np.array([200.0, 1e-9, 3.0], dtype=np.float32).view(np.uint8)
    r[3, 27] = 0
    rows = r.reshape(-1)
    sc = gh.Scene()
    events = []
    sc.addEventListener("change", lambda e: events.append(e))
    sc.setData(rows)
    data, pos = oracle.scene_pack(rows)
    assert events and sc.vertexCount == 20000
    assert sc.width == 2048 and sc.height == math.ceil(2 * 20000 / 2048)
    assert sc.data.size == sc.width * sc.height * 4
    assert np.array_equal(sc.positions, pos)
    assert np.array_equal(sc.data[:data.size], data)
    assert not sc.data[data.size:].any()
    assert not data.reshape(-1, 8)[:, 3].any()           # word 3 is never written (Scene.ts)


def test_camera_update_against_independent_formula():
    import gsplat_hip as gh
    for k in (0, 31, 77):
        cam = gh.orbit_camera(k, width=1920, height=1080)
        P = np.array(cam.projectionMatrix).reshape(4, 4).T   # buffer[c*4+r]
        V = np.array(cam.viewMatrix).reshape(4, 4).T
        VP = np.array(cam.viewProj).reshape(4, 4).T
        assert np.allclose(VP, P @ V, rtol=0, atol=1e-12)
        # view = [R^T | -R^T t]: the camera position maps to the origin, forward axis to +z
        t = np.array(cam.position)
        assert np.allclose(V @ np.append(t, 1.0), [0, 0, 0, 1], atol=1e-12)
        to_origin = V @ np.array([0.0, 0.0, 0.0, 1.0])
        assert abs(to_origin[0]) < 1e-9 and abs(to_origin[1]) < 1e-9 and abs(to_origin[2] - 8.0) < 1e-9
        assert P[0, 0] == 2 * 1132 / 1920 and P[1, 1] == -2 * 1132 / 1080 and P[3, 2] == 1
        R = V[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_band_edges_cover_the_image():
    from gsplat_hip import bands
    for W in (640, 1920, 3840, 333):
        for world in (1, 2, 3, 4, 8):
            e = bands.band_edges(W, world)
            assert e[0][0] == 0 and e[-1][1] == W
            for (a, b), (c, d) in zip(e[:-1], e[1:]):
                assert b == c and (a % 32 == 0 or a == W)
            assert all(b - a <= bands.slab_width(W, world) for a, b in e)


def test_scene_sh_packing_matches_oracle(oracle):
    import gsplat_hip as gh
    n, first = 600, 150
    rows = gh.synth.synth_rows(n, 5)
    rng = np.random.default_rng(3)
    shs = (rng.standard_normal((n - first, 48)) * 0.4).astype(np.float32)
    sc = gh.Scene()
    sc.bandsIndices = np.array([first - 1, 300, 450], dtype=np.int32)
    sc.setData(rows, shs)
    want = oracle.scene_pack_sh(shs)
    assert sc.shHeight == math.ceil(2 * (n - first) / 2048)
    for c in range(3):
        assert sc.shs_rgb[c].size == 2048 * sc.shHeight * 4
        assert np.array_equal(sc.shs_rgb[c][:want[c].size], want[c])


def test_balanced_band_edges():
    from gsplat_hip import bands
    rng = np.random.default_rng(0)
    for W in (640, 1920, 3840, 333):
        nbx = -(-W // 32)
        x = np.arange(nbx)
        cost = np.exp(-0.5 * ((x - nbx / 2) / (nbx / 6)) ** 2) * 1000 + 20 + rng.random(nbx)   # centre-heavy
        for world in (1, 2, 3, 4, 8):
            e = bands.balanced_edges(W, world, cost)
            assert len(e) == world and e[0][0] == 0 and e[-1][1] == W
            for (a, b), (c, d) in zip(e[:-1], e[1:]):
                assert b == c and (a % 32 == 0 or a == W)
            if world <= nbx:
                assert all(b > a for a, b in e)
                per = [cost[a // 32:-(-b // 32)].sum() for a, b in e]
                eq = [cost[a // 32:-(-b // 32)].sum() for a, b in bands.band_edges(W, world)]
                assert max(per) <= max(eq) + 1e-9               # never worse than equal-width bands
            assert all(b - a <= bands.slab_width(W, world, e) for a, b in e)
    assert bands.balanced_edges(64, 4, [1, 1]) == [(0, 32), (32, 64), (64, 64), (64, 64)]
