"""D1 / D2 of SURVEY 8(a) pinned by the reference ITSELF: tests/golden/host_golden.json holds inputs and outputs of
the reference's own TypeScript classes (Camera.update Camera.ts:32-56,81-92; Matrix4.multiply Matrix4.ts:32-53;
Matrix3.RotationFromQuaternion / multiply Matrix3.ts:33-80; Quaternion.multiply / normalize; Scene.setData with the SH
packing Scene.ts:58-180; translate / rotate / scale / limitBox Scene.ts:182-366), produced by
tests/golden/make_golden_host.js, which strips the type annotations at generation time and evaluates the classes under
Node.  Every producer of the buffers the hot path reads must equal them bit for bit:
the JavaScript host (gsplat.js_amd/js), the Python harness mirror, oracle.c's restatement and, on the GPU, k_scene.hip."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden", "host_golden.json")


@pytest.fixture(scope="module")
def g():
    return json.load(open(GOLDEN))


def f64(hexes):
    return [struct.unpack(">d", bytes.fromhex(h))[0] for h in hexes]


def arr(h, dtype):
    return np.frombuffer(bytes.fromhex(h), dtype=dtype)


def bits(xs):
    return [struct.pack(">d", float(x)).hex() for x in xs]


def test_fixture_was_made_by_running_the_reference(g):
    assert g["generator"] == "tests/golden/make_golden_host.js" and "evaluated under node" in g["source"]
    assert len(g["cameras"]) == 12 and len(g["quaternions"]) == 24 and g["scene"]["after_setData"]["vertexCount"] == 160
    assert 0 < g["scene"]["limitBox"]["after"]["vertexCount"] < 160
    assert g["scene"]["change_events"] == 5          # setData and the four transforms each dispatch "change"
    assert len(g["orbit"]) == 10 and all(len(o["steps"]) == 4 for o in g["orbit"])   # the reference's OrbitControls, executed


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_host_equals_the_reference_executed_goldens():
    r = subprocess.run([shutil.which("node"), os.path.join(HERE, "js", "host_check.js"), "golden", GOLDEN],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout)
    assert res["mismatches"] == [] and res["compared"] >= 240     # (incl. OrbitControls: 10 cases x constructor, update, setCameraTarget, 3 damped updates)


def test_python_mirror_camera_and_matrices(g):
    from gsplat_hip import camera as C
    for e in g["quaternions"]:
        q = f64(e["q"])
        assert bits(C.rotation_from_quaternion(*q)) == e["rotation_raw"]
        assert bits(C.rotation_from_quaternion(*f64(e["normalized"]))) == e["rotation_of_normalized"]
    for e in g["matrix4_products"]:
        assert bits(C.mat4_multiply(f64(e["a"]), f64(e["b"]))) == e["a_multiply_b"]
    for e in g["cameras"]:
        fx, fy, near, far = f64([e["fx"], e["fy"], e["near"], e["far"]])
        cam = C.Camera(f64(e["position"]), f64(e["rotation"]), fx, fy, near, far).update(e["width"], e["height"])
        assert bits(cam.projectionMatrix) == e["projectionMatrix"]
        assert bits(cam.viewMatrix) == e["viewMatrix"]
        assert bits(cam.viewProj) == e["viewProj"]


def test_python_mirror_scene_setdata(g):
    import gsplat_hip as gh
    rows = arr(g["scene"]["rows"], np.uint8)
    w = g["scene"]["after_setData"]
    sc = gh.Scene()
    sc.setData(rows)
    n = w["vertexCount"]
    assert sc.vertexCount == n and [sc.width, sc.height, sc.data.size] == [w["width"], w["height"], w["data_length"]]
    assert np.array_equal(sc.data[:8 * n], arr(w["data"], np.uint32)) and not sc.data[8 * n:].any()
    assert np.array_equal(sc.positions.view(np.uint32), arr(w["positions"], np.uint32))
    e = g["scene_sh"]
    shs = arr(e["shs"], np.float32)
    sc = gh.Scene()
    sc.bandsIndices = np.array([e["first"] - 1, 80, 120], dtype=np.int32)
    sc.setData(rows, shs)
    nsh = shs.size // 48
    assert sc.shHeight == e["shHeight"] and [t.size for t in sc.shs_rgb] == e["texture_words"]
    for c in range(3):
        assert np.array_equal(sc.shs_rgb[c][:8 * nsh], arr(e["shs_rgb"][c], np.uint32))


def _check_state(state, w, what):
    data, pos, rot, scl = state
    n = w["vertexCount"]
    assert np.array_equal(np.asarray(data)[:8 * n], arr(w["data"], np.uint32)), what + ": data"
    assert np.array_equal(np.asarray(pos)[:3 * n].view(np.uint32), arr(w["positions"], np.uint32)), what + ": positions"
    assert np.array_equal(np.asarray(rot)[:4 * n].view(np.uint32), arr(w["rotations"], np.uint32)), what + ": rotations"
    assert np.array_equal(np.asarray(scl)[:3 * n].view(np.uint32), arr(w["scales"], np.uint32)), what + ": scales"


def test_oracle_scene_restatement_equals_the_reference(g, oracle):
    s = g["scene"]
    rows = arr(s["rows"], np.uint8)
    data, pos = oracle.scene_pack(rows)
    assert np.array_equal(data, arr(s["after_setData"]["data"], np.uint32))
    assert np.array_equal(pos.view(np.uint32), arr(s["after_setData"]["positions"], np.uint32))
    st = oracle.SceneState(rows)
    get = lambda: (st.data, st.positions, st.rotations, st.scales)
    _check_state(get(), s["after_setData"], "setData")
    st.translate(f64(s["translate"]["t"]))
    _check_state(get(), s["translate"]["after"], "translate")
    st.rotate(f64(s["rotate"]["q"]))
    _check_state(get(), s["rotate"]["after"], "rotate")
    st.scale(f64(s["scale"]["s"]))
    _check_state(get(), s["scale"]["after"], "scale")
    st.limit_box(f64(s["limitBox"]["box"]))
    assert st.n == s["limitBox"]["after"]["vertexCount"]
    _check_state(get(), s["limitBox"]["after"], "limitBox")
    e = g["scene_sh"]
    tex = oracle.scene_pack_sh(arr(e["shs"], np.float32))
    for c in range(3):
        assert np.array_equal(tex[c], arr(e["shs_rgb"][c], np.uint32))


@pytest.mark.gpu
def test_device_scene_kernels_equal_the_reference(g):
    """k_scene.hip (gsr_set_scene_rows and the four transforms) against what the reference's Scene computed."""
    import gsplat_hip as gh
    s = g["scene"]
    r = gh.HIPRenderer(640, 480)
    try:
        r.set_scene_rows(arr(s["rows"], np.uint8))
        _check_state(r.read_scene(), s["after_setData"], "gsr_set_scene_rows")
        r.scene_translate(f64(s["translate"]["t"]))
        _check_state(r.read_scene(), s["translate"]["after"], "gsr_scene_translate")
        r.scene_rotate(f64(s["rotate"]["q"]))
        _check_state(r.read_scene(), s["rotate"]["after"], "gsr_scene_rotate")
        r.scene_scale(f64(s["scale"]["s"]))
        _check_state(r.read_scene(), s["scale"]["after"], "gsr_scene_scale")
        assert r.scene_limit_box(f64(s["limitBox"]["box"])) == s["limitBox"]["after"]["vertexCount"]
        _check_state(r.read_scene(), s["limitBox"]["after"], "gsr_scene_limit_box")
    finally:
        r.dispose()


# ---------------------------------------------------------------------------------------------------------------------
# (f) rank 4: the PLY loaders against the reference's own parsers, executed (tests/golden/make_golden_ply.py / .js)
# ---------------------------------------------------------------------------------------------------------------------
PLY_GOLDEN = os.path.join(HERE, "golden", "ply_golden.json")


@pytest.fixture(scope="module")
def pg():
    return json.load(open(PLY_GOLDEN))


def _rows_close(got, want, n):
    """exp / sigmoid go through libm here and through V8 in the reference: scales within an ulp, alpha / rotation bytes within 1."""
    got, want = got.reshape(n, 32), want.reshape(n, 32)
    assert np.array_equal(got[:, 0:12], want[:, 0:12])
    gs, ws = got[:, 12:24].copy().view(np.float32), want[:, 12:24].copy().view(np.float32)
    assert np.all(np.abs(gs - ws) <= np.spacing(np.abs(ws)))
    assert np.array_equal(got[:, 24:27], want[:, 24:27])
    assert np.abs(got[:, 27].astype(int) - want[:, 27].astype(int)).max() <= 1
    assert np.abs(got[:, 28:32].astype(int) - want[:, 28:32].astype(int)).max() <= 1


def test_ply_fixture_was_made_by_running_the_reference_parsers(pg):
    assert "evaluated under node" in pg["source"] and pg["q_bands"] == [39, 63, 79]
    assert len(pg["plain"]) == len(pg["polycam"]) == len(pg["full_rows"]) == 96 * 64 and len(pg["q_rows"]) == 100 * 64
    assert pg["plain"] != pg["polycam"]      # the axis swap


def test_python_ply_restatement_equals_the_reference_parsers(pg):
    from oracle import ply_oracle as P
    inria, quant = bytes.fromhex(pg["inria_ply"]), bytes.fromhex(pg["quantized_ply"])
    _rows_close(P.rows_from_ply(inria), arr(pg["plain"], np.uint8), 96)
    _rows_close(P.rows_from_ply(inria, "polycam"), arr(pg["polycam"], np.uint8), 96)
    rows, sh = P.rows_and_sh_from_ply(inria)
    _rows_close(rows, arr(pg["full_rows"], np.uint8), 96)
    assert np.array_equal(np.asarray(sh, dtype=np.float32).view(np.uint32), arr(pg["full_shs"], np.uint32))
    rows, sh, bands = P.rows_sh_from_qply(quant)
    _rows_close(rows, arr(pg["q_rows"], np.uint8), 100)
    assert np.array_equal(np.asarray(sh, dtype=np.float32).view(np.uint32), arr(pg["q_shs"], np.uint32))
    assert list(bands) == pg["q_bands"]


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_ply_loader_equals_the_reference_parsers_bit_for_bit(pg, tmp_path):
    """Same engine on both sides (V8's Math.exp): rows, SH floats and bandsIndices must be identical."""
    node = shutil.which("node")
    a, q = tmp_path / "a.ply", tmp_path / "q.ply"
    a.write_bytes(bytes.fromhex(pg["inria_ply"]))
    q.write_bytes(bytes.fromhex(pg["quantized_ply"]))
    r = subprocess.run([node, os.path.join(HERE, "js", "host_check.js"), "plygolden", str(a), str(q)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    for k in ("plain", "polycam", "full_rows", "full_shs", "q_rows", "q_shs", "q_bands"):
        assert got[k] == pg[k], k
