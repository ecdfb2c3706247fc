"""The JavaScript host (gsplat.js_amd/js, the reference's own language) driven through Node.
CPU part: Scene.setData / transforms / Camera.update bits against the oracle and the Python mirror, the
export list of src/index.ts, and the loud failure without a GPU.  GPU part: renderer.render(scene, camera)
through the N-API addon against the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "js", "host_check.js")
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "gsplat.js_amd", "js", "native", "gsplat_hip.node")

pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def run(*args):
    r = subprocess.run([NODE, DRIVER] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_export_list_matches_reference_index():
    names = json.loads(run("api"))
    for n in ["Camera", "Scene", "Loader", "WebGLRenderer", "OrbitControls", "Quaternion", "Vector3", "Matrix4", "Matrix3",
              "ShaderPass", "FadeInPass"]:   # src/index.ts:1-12 minus PLYLoader (out of scope)
        assert n in names


def test_scene_and_camera_bits(tmp_path, oracle):
    import gsplat_hip as gh
    rows = gh.synth.synth_rows(5000, 123)
    f = tmp_path / "s.splat"
    rows.tofile(f)
    out = str(tmp_path / "o")
    poses = (0, 31, 77)
    run("pack", f, out, 1920, 1080, 1132, *poses)
    meta = json.load(open(out + ".json"))
    data, pos = oracle.scene_pack(rows)
    jdata = np.fromfile(out + ".data.bin", dtype=np.uint32)
    assert meta["events"] == 1 and meta["vertexCount"] == 5000 and meta["width"] == 2048
    assert jdata.size == meta["dataLength"] == 2048 * meta["height"] * 4
    assert np.array_equal(jdata[:data.size], data) and not jdata[data.size:].any()
    assert np.array_equal(np.fromfile(out + ".pos.bin", dtype=np.float32), pos)
    # Camera.update: JS f64 matrices equal the Python mirror's (same formulas, same order) to the last bit,
    # except the trigonometric pose inputs, which may differ by an ulp between V8 and libm
    for cam_js in meta["cams"]:
        cam = gh.Camera(tuple(cam_js["position"]), tuple(cam_js["rotation"])).update(1920, 1080)
        assert cam.viewMatrix == cam_js["view"]
        assert cam.projectionMatrix == cam_js["proj"]
        assert cam.viewProj == cam_js["viewProj"]
        ref = gh.orbit_camera(cam_js["pose"])
        assert np.allclose(ref.viewProj, cam_js["viewProj"], rtol=0, atol=1e-12)
    # transforms (Scene.ts:182-366) bit-exact against the oracle's f64 restatement, same order of operations
    st = oracle.SceneState(rows)
    st.translate([0.25, -0.5, 1.0])
    st.rotate(meta["q"])                      # the quaternion JS computed (V8 trigonometry), passed through exactly
    st.scale([1.5, 1.5, 1.5])
    st.limit_box([-4, 4, -4, 4, -4, 4])
    assert meta["xfCount"] == st.n and 0 < st.n < 5000
    assert np.array_equal(np.fromfile(out + ".xf.pos.bin", dtype=np.float32), st.positions)
    assert np.array_equal(np.fromfile(out + ".xf.rot.bin", dtype=np.float32), st.rotations)
    assert np.array_equal(np.fromfile(out + ".xf.scl.bin", dtype=np.float32), st.scales)
    assert np.array_equal(np.fromfile(out + ".xf.data.bin", dtype=np.uint32)[:8 * st.n], st.data)
    assert np.fromfile(out + ".xf.splat", dtype=np.uint8).size == st.n * 32


def test_js_scene_sh_packing(tmp_path, oracle):
    import gsplat_hip as gh
    n, first = 500, 120
    rows = gh.synth.synth_rows(n, 9)
    shs = (np.random.default_rng(4).standard_normal((n - first, 48)) * 0.4).astype(np.float32)
    f, g = tmp_path / "s.splat", tmp_path / "s.shs"
    rows.tofile(f)
    shs.tofile(g)
    out = str(tmp_path / "o")
    run("packsh", f, g, out, first - 1, 250, 400)
    want = oracle.scene_pack_sh(shs)
    meta = json.load(open(out + ".json"))
    assert meta["shHeight"] == -(-(2 * (n - first)) // 2048)
    for c in range(3):
        got = np.fromfile(out + ".sh%d.bin" % c, dtype=np.uint32)
        assert np.array_equal(got[:want[c].size], want[c]) and not got[want[c].size:].any()


def test_js_ply_loader(tmp_path, oracle):
    # INRIA .ply -> .splat rows (+ SH floats), against a numpy restatement of PLYLoader.ts
    from oracle import ply_oracle as P
    n = 3000
    ply = P.synth_ply(n, 17)
    f = tmp_path / "s.ply"
    f.write_bytes(ply)
    out = str(tmp_path / "p")
    run("ply", f, out)
    meta = json.load(open(out + ".json"))
    assert meta == {"n": n, "refused": True, "badMagic": True}

    def check_rows(tag, want):
        got = np.fromfile(out + "." + tag + ".splat", dtype=np.uint8).reshape(n, 32)
        want = want.reshape(n, 32)
        assert np.array_equal(got[:, 0:12], want[:, 0:12])                                    # positions
        gs, ws = got[:, 12:24].copy().view(np.float32), want[:, 12:24].copy().view(np.float32)
        assert np.all(np.abs(gs - ws) <= np.spacing(ws))                                      # exp: V8 vs libm, <= 1 ulp
        assert np.array_equal(got[:, 24:27], want[:, 24:27])                                  # colour bytes
        assert np.abs(got[:, 27].astype(int) - want[:, 27].astype(int)).max() <= 1            # sigmoid byte
        assert np.abs(got[:, 28:32].astype(int) - want[:, 28:32].astype(int)).max() <= 1      # rotation bytes
        assert (got[:, 28:32] == want[:, 28:32]).mean() > 0.99

    check_rows("plain", P.rows_from_ply(ply))
    check_rows("polycam", P.rows_from_ply(ply, "polycam"))
    rows_full, sh = P.rows_and_sh_from_ply(ply)
    check_rows("full", rows_full)
    want_sh = oracle.scene_pack_sh(sh)          # bandsIndices (-1,-1,-1): every splat carries degree-3 SH
    for c in range(3):
        got = np.fromfile(out + ".full.sh%d.bin" % c, dtype=np.uint32)
        assert np.array_equal(got[:want_sh[c].size], want_sh[c])
    assert not np.array_equal(rows_full.reshape(n, 32)[:, 24:27], P.rows_from_ply(ply).reshape(n, 32)[:, 24:27])   # the precedence oddity is real


def test_js_quantized_ply_loader(tmp_path, oracle):
    # SURVEY 8(f) rank 4, codebook variant (PLYLoader.ts:893-1196): four vertex elements with 0..3 SH bands, half
    # positions, one-byte codebook indices; rows, SH floats and bandsIndices against the numpy restatement
    from oracle import ply_oracle as P
    counts = [700, 500, 300, 400]
    ply = P.synth_qply(counts, 23)
    f = tmp_path / "q.ply"
    f.write_bytes(ply)
    out = str(tmp_path / "q")
    run("qply", f, out)
    n = sum(counts)
    want_rows, want_sh, want_bands = P.rows_sh_from_qply(ply)
    meta = json.load(open(out + ".json"))
    assert meta["n"] == n and meta["bands"] == list(want_bands) == [699, 1199, 1499] and meta["parsedBands"] == meta["bands"]
    got = np.fromfile(out + ".rows.bin", dtype=np.uint8).reshape(n, 32)
    want = want_rows.reshape(n, 32)
    assert np.array_equal(got[:, 0:12], want[:, 0:12])                                    # half positions, exact
    gs, ws = got[:, 12:24].copy().view(np.float32), want[:, 12:24].copy().view(np.float32)
    assert np.all(np.abs(gs - ws) <= np.spacing(ws))                                      # exp: V8 vs libm, <= 1 ulp
    assert np.array_equal(got[:, 24:27], want[:, 24:27])
    assert np.abs(got[:, 27].astype(int) - want[:, 27].astype(int)).max() <= 1
    assert np.abs(got[:, 28:32].astype(int) - want[:, 28:32].astype(int)).max() <= 1
    got_sh = np.fromfile(out + ".shs.bin", dtype=np.float32)
    assert got_sh.size == 48 * (n - counts[0]) and np.array_equal(got_sh.view(np.uint32), want_sh.view(np.uint32))
    sh2 = got_sh.reshape(-1, 48)
    assert not sh2[:counts[1], 12:].any() and sh2[:counts[1], 3:12].any()                 # one band: 9 higher coefficients, rest zero
    assert not sh2[counts[1]:counts[1] + counts[2], 27:].any()                           # two bands: 24
    assert sh2[counts[1] + counts[2]:, 27:].any()                                         # three bands: all 45
    # and through Scene.setData: the three half textures hold exactly the SH-carrying splats (i > bandsIndices[0])
    want_tex = oracle.scene_pack_sh(want_sh)
    for c in range(3):
        tex = np.fromfile(out + ".sh%d.bin" % c, dtype=np.uint32)
        assert np.array_equal(tex[:want_tex[c].size], want_tex[c]) and not tex[want_tex[c].size:].any()
    assert np.array_equal(np.fromfile(out + ".splat", dtype=np.uint8).reshape(n, 32)[:, 0:12], want[:, 0:12])


def test_js_sort_baseline_matches_oracle(tmp_path, oracle):
    # tools/sort_js_baseline.js (the CPU stand-in bench.py times beside the native reference sort) is the same algorithm:
    # its depthIndex hashes to the oracle's
    import gsplat_hip as gh
    rows = gh.synth.synth_rows(50000, 9)
    data, pos = oracle.scene_pack(rows)
    f = tmp_path / "pos.f32"
    np.asarray(pos, dtype=np.float32).tofile(f)
    vp = gh.orbit_camera(31).f32()[2]
    r = subprocess.run(["node", os.path.join(ROOT, "tools", "sort_js_baseline.js"), str(f), repr(float(vp[2])), repr(float(vp[6])),
                        repr(float(vp[10])), "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    di, _, _ = oracle.sort(vp, pos)
    h = 0xcbf29ce484222325
    for b in di.astype("<u4").tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    assert got["n"] == 50000 and int(got["checksum"], 16) == h


def test_js_renderer_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(ADDON):
        pytest.skip("addon not built")
    out = run("nodevice")
    assert out.startswith("THROWN:") and "no HIP device" in out


@pytest.mark.gpu
def test_js_frames_in_flight(tmp_path):
    # Node host, three `throughput` renderers used round-robin through renderAsync()/sync(): every frame equals the one a
    # default renderer draws synchronously (permutation bit for bit, RGBA8 within one step: the segment lengths differ)
    import gsplat_hip as gh
    cfg = gh.synth.CONFIGS["C1"]
    f = tmp_path / "c1.splat"
    gh.synth.config_rows("C1").tofile(f)
    out = str(tmp_path / "f")
    run("inflight", f, out, cfg["width"], cfg["height"], cfg["fx"])
    assert json.load(open(out + ".json")) == {"same": True, "frames": 12}


@pytest.mark.gpu
def test_js_render_with_in_library_allgather(tmp_path):
    # renderer.render(scene, camera) on a renderer that has joined a group: band frame + RCCL all-gather of the RGBA8
    # slabs inside the library, no torch anywhere in the process.  One rank is what a one-GPU box can run; the gathered
    # frame must be the plain renderer's frame (same kernels, RGBA8 conversion in the pack kernel instead of k_to_rgba8).
    import gsplat_hip as gh
    cfg = gh.synth.CONFIGS["C1"]
    f = tmp_path / "c1.splat"
    gh.synth.config_rows("C1").tofile(f)
    out = str(tmp_path / "g")
    run("group", f, out, cfg["width"], cfg["height"], cfg["fx"], 0, 1, tmp_path / "comm.id")
    got = json.load(open(out + ".json"))
    assert got["worst"] == 0 and got["world"] == 1 and got["group"] == {"rank": 0, "world": 1}
    # and with a second renderer sharing the group (renderer.shareGroup: frames in flight of one rank)
    assert got["worstShared"] == 0 and got["group2"] == {"rank": 0, "world": 1}


@pytest.mark.gpu
def test_js_render_matches_oracle(tmp_path, oracle):
    import gsplat_hip as gh
    cfg = gh.synth.CONFIGS["C1"]
    rows = gh.synth.config_rows("C1")
    f = tmp_path / "c1.splat"
    rows.tofile(f)
    out = str(tmp_path / "r")
    run("render", f, out, cfg["width"], cfg["height"], cfg["fx"], 9)
    meta = json.load(open(out + ".json"))
    data, pos = oracle.scene_pack(rows)
    # the exact camera the JS host used (its f64 viewProj), rounded to f32 like Float32Array(buffer)
    vp = np.asarray(meta["viewProj"], dtype=np.float32)
    odi, _, _ = oracle.sort(vp, pos)
    assert np.array_equal(np.fromfile(out + ".depthIndex.bin", dtype=np.uint32), odi)
    cam = gh.orbit_camera(9, width=cfg["width"], height=cfg["height"], fx=cfg["fx"])
    v, p, vp2 = cam.f32()
    oimg = oracle.render_scene(data, pos, v, p, vp2, cam.fx, cam.fy, cfg["width"], cfg["height"], mode=1)[0]
    img = np.fromfile(out + ".rgba32f.bin", dtype=np.float32).reshape(cfg["height"], cfg["width"], 4)
    # V8's and libm's sin/cos may differ by an ulp in the pose, hence a slightly wider bound than the 2e-4 of the
    # Python-driven parity tests (which use bit-identical cameras on both sides)
    assert np.abs(img.astype(np.float64) - oimg).max() <= 1e-3
    img8 = np.fromfile(out + ".rgba8.bin", dtype=np.uint8).reshape(cfg["height"], cfg["width"], 4)
    o8 = np.floor(np.clip(oimg.astype(np.float64), 0, 1) * 255.0 + 0.5).astype(np.int32)
    assert np.abs(img8.astype(np.int32) - o8).max() <= 1
    # scene.translate fired "change": the renderer re-uploaded and sorted the moved positions
    moved = np.fromfile(out + ".moved.pos.bin", dtype=np.float32)
    mdi = oracle.sort(vp, moved)[0]
    assert np.array_equal(np.fromfile(out + ".moved.depthIndex.bin", dtype=np.uint32), mdi)
    # the wasm export's drop-in (7-argument sort on host arrays), called after the move
    assert np.array_equal(np.fromfile(out + ".sortHost.depthIndex.bin", dtype=np.uint32), mdi)
    assert meta["stats"]["n"] == cfg["n"] and meta["device"]["computeUnits"] > 0


@pytest.mark.gpu
def test_js_default_fade_in_pass(tmp_path, oracle):
    # new WebGLRenderer() installs a FadeInPass (WebGLRenderer.ts:41-44): frame k is drawn with u_depthFade = 0.01 k
    import gsplat_hip as gh
    cfg = gh.synth.CONFIGS["C1"]
    rows = gh.synth.config_rows("C1")
    f = tmp_path / "c1.splat"
    rows.tofile(f)
    out = str(tmp_path / "f")
    run("renderfade", f, out, cfg["width"], cfg["height"], cfg["fx"], 9, 30)
    data, pos = oracle.scene_pack(rows)
    cam = gh.orbit_camera(9, width=cfg["width"], height=cfg["height"], fx=cfg["fx"])
    v, p, vp = cam.f32()
    fade = 0.0
    for _ in range(30):
        fade = min(fade + 0.01, 1.0)
    oimg = oracle.render_scene(data, pos, v, p, vp, cam.fx, cam.fy, cfg["width"], cfg["height"], mode=1, fade=fade)[0]
    img = np.fromfile(out + ".rgba32f.bin", dtype=np.float32).reshape(cfg["height"], cfg["width"], 4)
    assert oimg[..., 3].max() > 0.05
    assert np.abs(img.astype(np.float64) - oimg).max() <= 2e-3   # pose trigonometry differs by an ulp between V8 and libm


@pytest.mark.gpu
def test_js_device_scene_matches_js_scene(tmp_path):
    # device-side Scene.setData + translate/rotate/scale/limitBox (kernels) against the JavaScript Scene: identical words,
    # identical pixels
    import gsplat_hip as gh
    rows = gh.synth.synth_rows(20000, 77)
    f = tmp_path / "s.splat"
    rows.tofile(f)
    out = str(tmp_path / "d")
    run("devscene", f, out)
    meta = json.load(open(out + ".json"))
    assert meta["same"] and meta["samePixels"] and 0 < meta["n"] < 20000
