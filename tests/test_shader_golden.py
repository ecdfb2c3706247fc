"""The render half of the oracle against the reference's OWN shaders, executed: tests/golden/shader_golden.json holds inputs
and outputs of vertex.glsl.ts / frag.glsl.ts, produced by tests/golden/make_golden_shader.py, which translates the GLSL text
into Python at generation time (tests/golden/glsl_eval.py: IEEE binary32 operations, the evaluation rules DESIGN.md 4
states) and runs main() for every sample splat and corner, sample fragments, and eval_sh_rgb on sample directions.
oracle.c's orc_project* / fragment weight / SH polynomial must reproduce what the shader computed: the varyings and the
quad's axes bit for bit (projection and SH: f32 both sides), the fragment colour to f32 rounding (the oracle weighs in f64).
What this pins is the STRUCTURE of the restatement -- which operations on which operands in which order -- by execution
instead of by reading; what no test here can pin is a GPU's own rounding of GLSL divisions and square roots."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "shader_golden.json")
F = np.float32


@pytest.fixture(scope="module")
def g():
    return json.load(open(GOLDEN))


def f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def test_fixture_was_made_by_running_the_reference_shaders(g):
    assert g["generator"] == "tests/golden/make_golden_shader.py" and "executed" in g["source"]
    assert len(g["cameras"]) == 6 and all(len(c["splats"]) == 160 for c in g["cameras"])
    assert len(g["fragments"]) == 400 and len(g["sh"]["cases"]) == 96
    # every exit of the vertex shader is taken by some sample: frustum cull, a return before the axes, NaN axes, a drawn quad
    kinds = set()
    for c in g["cameras"]:
        for s in c["splats"]:
            gp = s["gl_Position"][0]
            if gp is None:
                kinds.add("returned")
            elif list(f32(gp)) == [0.0, 0.0, 2.0, 1.0] and s["majorAxis"] is None:
                kinds.add("culled")
            elif not np.isfinite(f32(s["majorAxis"])).all():
                kinds.add("nan axes")
            else:
                kinds.add("drawn")
    assert {"culled", "nan axes", "drawn"} <= kinds, kinds


def test_projection_equals_the_executed_vertex_shader(g, oracle):
    data = np.frombuffer(bytes.fromhex(g["data_words"]), dtype=np.uint32)
    W, H = g["width"], g["height"]
    drawn = 0
    for c in g["cameras"]:
        view, proj = f32(c["view"]), f32(c["projection"])
        fade = float(f32([c["fade"]])[0]) if c["use_fade"] else None
        rec, bbox, raw = oracle.project(data, view, proj, c["fx"], c["fy"], W, H, fade=fade)
        for i, s in enumerate(c["splats"]):
            what = (c["use_fade"], i)
            gp0 = s["gl_Position"][0]
            visible = raw[i, 11] != 0.0
            if gp0 is None or s["majorAxis"] is None:            # culled, or returned before it placed the vertex
                assert not visible, what
                continue
            maj, mnr, vc = f32(s["majorAxis"]), f32(s["minorAxis"]), f32(s["vCenter"])
            sf = f32([s["scalingFactor"]])[0]
            with np.errstate(all="ignore"):
                axes = np.array([maj[0] * sf, maj[1] * sf, mnr[0] * sf, mnr[1] * sf], dtype=np.float32)
            if not visible:
                # the shader went through; nothing is drawn all the same: NaN axes (normalize(0, 0)), a quad scaled to zero by the
                # depth fade, or an axis of length zero (such a splat is dropped, as the reference loses it)
                m2 = axes[0] * axes[0] + axes[1] * axes[1]
                n2 = axes[2] * axes[2] + axes[3] * axes[3]
                assert (not np.isfinite(axes).all()) or not (sf > 0) or m2 == 0 or n2 == 0 or not np.isfinite([2 / m2, 2 / n2]).all(), what
                continue
            drawn += 1
            assert np.array_equal(raw[i, 2:6].view(np.uint32), axes.view(np.uint32)), (what, raw[i, 2:6], axes)
            col = f32(s["vColor"])
            assert np.array_equal(raw[i, 6:10].view(np.uint32), col[[3, 0, 1, 2]].view(np.uint32)), what
            assert raw[i, 10].view(np.uint32) == np.uint32(s["pos2d_w"]), what
            xw = ((vc[0] + F(1.0)) * F(0.5)) * F(W)
            yw = ((vc[1] + F(1.0)) * F(0.5)) * F(H)
            assert raw[i, 0] == xw and raw[i, 1] == yw, what
            # and the vertex positions themselves: gl_Position.xy = vCenter + position.x * major * s / viewport + position.y * minor * s / viewport
            for (px, py), gp in zip([(-2.0, -2.0), (2.0, -2.0), (2.0, 2.0), (-2.0, 2.0)], s["gl_Position"]):
                got = f32(gp)
                ex = (vc[0] + (F(px) * raw[i, 2]) / F(W)) + (F(py) * raw[i, 4]) / F(W)
                ey = (vc[1] + (F(px) * raw[i, 3]) / F(H)) + (F(py) * raw[i, 5]) / F(H)
                assert got[0] == ex and got[1] == ey and got[2] == 0.0 and got[3] == 1.0, what
    assert drawn > 400


def test_fragment_weight_equals_the_executed_fragment_shader(g, oracle):
    kept = 0
    for fr in g["fragments"]:
        v, col = f32(fr["vPosition"]), f32(fr["vColor"])
        got = oracle.fragment(v, col)
        q = float(v[0]) * float(v[0]) + float(v[1]) * float(v[1])
        if abs(q - 4.0) < 1e-5 and (got is None) != (fr["fragColor"] is None):
            continue                                  # (f32 and f64 may round |vPosition|^2 to different sides of 4)
        assert (got is None) == (fr["fragColor"] is None), (v, q)
        if got is not None:
            kept += 1
            assert np.abs(got.astype(np.float64) - f32(fr["fragColor"]).astype(np.float64)).max() <= 2.5e-7, (v, got, f32(fr["fragColor"]))
    assert kept > 250
    # the discard edge exactly: |vPosition|^2 == 4 is drawn, the next f32 above is not
    edge = [fr for fr in g["fragments"][:8]]
    assert any(fr["fragColor"] is None for fr in edge) and any(fr["fragColor"] is not None for fr in edge)


def test_sh_polynomial_equals_the_executed_eval_sh_rgb(g, oracle):
    sh = [np.frombuffer(bytes.fromhex(w), dtype=np.uint32) for w in g["sh"]["words"]]
    # (the textures themselves: Scene.setData's half packing of the seeded coefficients, pinned by tests/test_host_golden.py)
    packed = oracle.scene_pack_sh(np.frombuffer(bytes.fromhex(g["sh"]["shs"]), dtype=np.float32))
    for c in range(3):
        assert np.array_equal(packed[c][:sh[c].size], sh[c])
    for case in g["sh"]["cases"]:
        want = np.minimum(f32(case["rgb"]), F(1.0))          # (main() clamps what eval_sh_rgb returns: vertex.glsl.ts:200)
        got = oracle.eval_sh_rgb(sh, case["index"], case["deg"], f32(case["dir"]))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case["index"], case["deg"], got, want)


@pytest.mark.gpu
def test_device_projection_equals_the_executed_vertex_shader(g, oracle):
    """k_project_key on the fixture's scene and cameras (incl. the depth fade and the unrotated camera that produces NaN axes):
    the axes and colours the executed shader computed, turned into this build's record by the oracle's own last step, are what
    the kernel writes -- bit for bit."""
    import gsplat_hip as gh
    rows = np.frombuffer(bytes.fromhex(g["rows"]), dtype=np.uint8)
    data = np.frombuffer(bytes.fromhex(g["data_words"]), dtype=np.uint32)
    W, H = g["width"], g["height"]
    r = gh.HIPRenderer(W, H)
    r.set_scene_rows(rows)
    drawn = 0
    try:
        for c in g["cameras"]:
            view, proj = f32(c["view"]), f32(c["projection"])
            fade = float(f32([c["fade"]])[0])
            r.set_depth_fade(c["use_fade"], fade if c["use_fade"] else 1.0)
            vp = np.zeros(16, dtype=np.float32)      # (the sort key's matrix: not under test here)
            vp[[2, 6, 10, 14]] = [0.0, 0.0, 1.0, 0.0]
            r.set_camera_arrays(view, proj, vp, c["fx"], c["fy"])
            r.render_async(); r.sync()
            rec, bbox = r.read_records()
            orec, obbox, oraw = oracle.project(data, view, proj, c["fx"], c["fy"], W, H, fade=fade if c["use_fade"] else None)
            assert np.array_equal(bbox, obbox)
            vis = oraw[:, 11] == 1.0
            drawn += int(vis.sum())
            for col in (0, 1, 2, 3, 4, 5, 7):
                assert np.array_equal(rec[vis][:, col].view(np.uint32), orec[vis][:, col].view(np.uint32)), col
        assert drawn > 400     # (one of the fade cameras scales every quad to nothing)
    finally:
        r.dispose()


def test_blend_state_read_from_the_renderer_and_applied(g, oracle):
    """C2: the fixed-function state the reference draws with, read from WebGLRenderer.ts at generation time, is the one the
    oracle (and the kernels) implement -- clear to (0, 0, 0, 0), one instanced TRIANGLE_FAN of 4 vertices at (+-2, +-2),
    FUNC_ADD with (ONE_MINUS_DST_ALPHA, ONE) for colour and alpha = front-to-back "under" on premultiplied fragments --
    and that state applied (OpenGL ES 3.0 blend equations, f32 destination) to sequences of the executed fragment
    shader's outputs gives what orc_render's per-pixel accumulation gives (f64: to f32 rounding)."""
    st = g["gl_state"]
    assert st["blendFuncSeparate"] == [["ONE_MINUS_DST_ALPHA", "ONE", "ONE_MINUS_DST_ALPHA", "ONE"]]
    assert st["blendEquationSeparate"] == [["FUNC_ADD", "FUNC_ADD"]] and st["clearColor"] == [["0", "0", "0", "0"]]
    assert st["drawArraysInstanced"] == [["TRIANGLE_FAN", "0", "4"]] and st["quad"] == [-2.0, -2.0, 2.0, -2.0, 2.0, 2.0, -2.0, 2.0]
    assert len(g["blend_sequences"]) == 24
    alphas = []
    for seq in g["blend_sequences"]:
        frags = np.array([f32(it) for it in seq["fragments"]], dtype=np.float32)
        got = oracle.composite(frags)
        want = f32(seq["dst"])
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 1e-6, (got, want)
        alphas.append(float(want[3]))
    assert max(alphas) > 0.5 and min(alphas) < 0.5       # thin and thick stacks both
