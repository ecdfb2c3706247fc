#!/usr/bin/env python3
"""tests/golden/shader_golden.json: inputs and outputs of the reference's OWN vertex and fragment shaders
(src/renderers/webgl/shaders/vertex.glsl.ts, frag.glsl.ts), executed from their text.

GLSL cannot run in this container (no GL, no GPU), and the reference ships no rendered images; until round 4 the render
half of the oracle (oracle.c: orc_project*, the fragment weight of orc_render) was pinned only by reading it against the
shader.  This script reads the two shader modules where they lie under /root/reference, translates the GLSL statement by
statement into Python (tests/golden/glsl_eval.py: a front end for the subset the shaders use and an IEEE-binary32 runtime
with the evaluation rules DESIGN.md 4 states) and RUNS main() -- every splat of the samples below through the vertex
shader for the four corners of its quad, sample fragments through the fragment shader, eval_sh_rgb on sample directions.
What is stored: the inputs (scene words, SH words, matrices, uniforms) and what the shader computed (its outputs and the
locals the oracle's `raw` record corresponds to).  Never the shader text.

Run here (the reference is not on the GPU box):  python tests/golden/make_golden_shader.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "gsplat.js_amd", "py")]
import glsl_eval as G            # noqa: E402
import gsplat_hip as gh          # noqa: E402

REF = "/root/reference/src/renderers/webgl/shaders"
F = np.float32


def hexf(x):
    return np.array([x], dtype=np.float32).view(np.uint32)[0].item()


def vec_bits(v):
    return [hexf(c) for c in v.c]


def sample_rows():
    """C1's first splats + hand-made ones that take the shader's early exits and its degenerate branches."""
    rows = np.array(gh.synth.synth_rows(160, 1, 1.5, 0.004, 0.06)).reshape(-1, 32).copy()

    def put(k, pos, scale, rgba, rot):
        rows[k, 0:12] = np.array(pos, dtype=np.float32).view(np.uint8)
        rows[k, 12:24] = np.array(scale, dtype=np.float32).view(np.uint8)
        rows[k, 24:28] = rgba
        rows[k, 28:32] = rot
    put(150, (0.0, 0.0, 0.0), (0.05, 0.05, 0.05), (255, 0, 0, 255), (255, 128, 128, 128))     # round: normalize(0, 0)
    put(151, (0.2, 0.1, 0.0), (0.30, 0.02, 0.02), (0, 255, 0, 200), (255, 128, 128, 128))     # axis aligned, long in x
    put(152, (0.0, 0.3, 0.1), (0.02, 0.30, 0.02), (0, 0, 255, 90), (255, 128, 128, 128))      # axis aligned, long in y
    put(153, (50.0, 0.0, 0.0), (0.05, 0.05, 0.05), (9, 9, 9, 255), (200, 90, 160, 40))        # far off screen: frustum cull
    put(154, (0.0, 0.0, 40.0), (0.05, 0.05, 0.05), (9, 9, 9, 255), (200, 90, 160, 40))        # behind some of the cameras
    put(155, (0.0, 0.0, 0.0), (1e-5, 1e-5, 1e-5), (255, 255, 255, 255), (255, 128, 128, 128)) # a point: the 0.3 blur only
    put(156, (0.5, -0.4, 0.2), (3.0, 3.0, 0.001), (120, 130, 140, 255), (180, 160, 100, 90))  # huge and flat: the 1024 clamp
    put(157, (0.1, 0.1, 0.1), (0.0, 0.0, 0.0), (1, 2, 3, 4), (255, 128, 128, 128))            # zero scale
    return rows.reshape(-1)


def main():
    vert_ns, _ = G.compile_shader(open(os.path.join(REF, "vertex.glsl.ts")).read())
    frag_ns, _ = G.compile_shader(open(os.path.join(REF, "frag.glsl.ts")).read())

    rows = sample_rows()
    scene = gh.Scene()
    scene.setData(rows)
    n = scene.vertexCount
    data = np.asarray(scene.data[:8 * n], dtype=np.uint32)
    W, H = 640, 480
    out = {"generator": "tests/golden/make_golden_shader.py",
           "source": "vertex.glsl.ts / frag.glsl.ts of the reference, translated by tests/golden/glsl_eval.py and executed (numpy.float32)",
           "width": W, "height": H, "rows": rows.tobytes().hex(), "data_words": data.tobytes().hex(), "cameras": []}

    tex = G.Texture([int(w) for w in scene.data], 2048)
    corners = [(-2.0, -2.0), (2.0, -2.0), (2.0, 2.0), (-2.0, 2.0)]
    cams = [(gh.orbit_camera(7, 120, W, H, 560.0), False, 1.0), (gh.orbit_camera(52, 120, W, H, 560.0), False, 1.0),
            (gh.orbit_camera(95, 120, W, H, 560.0, beta=0.9, radius=3.0), False, 1.0),
            (gh.orbit_camera(20, 120, W, H, 560.0), True, 0.02), (gh.orbit_camera(20, 120, W, H, 560.0), True, 0.0745),
            # no rotation, splat 151 on the optical axis: its footprint is axis aligned and longer in x, so the shader's
            # normalize(vec2(cov2d[0][1], lambda1 - cov2d[0][0])) is normalize(0, 0)
            (gh.camera.Camera((0.2, 0.1, -5.0), (0.0, 0.0, 0.0, 1.0), 560.0, 560.0).update(W, H), False, 1.0)]
    for cam, use_fade, fade in cams:
        v, p, vp = cam.f32()
        ns = vert_ns
        ns["u_texture"] = tex
        ns["view"] = ns["mat4"](*[F(x) for x in v])
        ns["projection"] = ns["mat4"](*[F(x) for x in p])
        ns["focal"] = ns["vec2"](F(560.0), F(560.0))
        ns["viewport"] = ns["vec2"](F(W), F(H))
        ns["u_useDepthFade"] = use_fade
        ns["u_depthFade"] = F(fade)
        ns["u_bandIndex"] = G.IVec([1 << 30, 1 << 30, 1 << 30])      # no splat has SH coefficients in this scene
        entry = {"view": [hexf(x) for x in v], "projection": [hexf(x) for x in p], "fx": 560.0, "fy": 560.0,
                 "use_fade": use_fade, "fade": hexf(F(fade)), "splats": []}
        with np.errstate(all="ignore"):
            for i in range(n):
                ns["index"] = i
                rec = {"gl_Position": []}
                for cx, cy in corners:
                    ns["position"] = ns["vec2"](F(cx), F(cy))
                    ns["gl_Position"] = ns["vColor"] = ns["vPosition"] = None
                    loc = ns["main"]()     # (the translated main() returns its locals: the shader's intermediate values)
                    rec["gl_Position"].append(None if ns["gl_Position"] is None else vec_bits(ns["gl_Position"]))
                for k in ("majorAxis", "minorAxis", "vCenter"):
                    rec[k] = vec_bits(loc[k]) if k in loc and loc[k] is not None else None
                rec["scalingFactor"] = hexf(loc["scalingFactor"]) if "scalingFactor" in loc else None
                rec["vColor"] = None if ns["vColor"] is None else vec_bits(ns["vColor"])
                rec["pos2d_w"] = hexf(loc["pos2d"].w)
                entry["splats"].append(rec)
        out["cameras"].append(entry)

    # fragments: the weight and the colour the fragment shader gives for sample varyings
    rng = np.random.default_rng(5)
    frags = []
    with np.errstate(all="ignore"):
        for k in range(400):
            r = 2.3 * np.sqrt(rng.random())
            a = 2.0 * np.pi * rng.random()
            vpos = (F(r * np.cos(a)), F(r * np.sin(a)))
            col = [F(x) for x in rng.random(4)]
            if k < 8:     # the discard edge: |vPosition|^2 just below, at and above 4
                vpos = (F(2.0) if k % 2 == 0 else np.nextafter(F(2.0), F(3.0 if k >= 4 else 0.0)), F(0.0))
            frag_ns["vPosition"] = frag_ns["vec2"](*vpos)
            frag_ns["vColor"] = frag_ns["vec4"](*col)
            frag_ns["fragColor"] = None
            try:
                frag_ns["main"]()
                res = vec_bits(frag_ns["fragColor"])
            except G.Discard:
                res = None
            frags.append({"vPosition": [hexf(x) for x in vpos], "vColor": [hexf(x) for x in col], "fragColor": res})
    out["fragments"] = frags

    # The fixed-function state the reference draws with, read from WebGLRenderer.ts (comments removed): blend factors and
    # equation, clear colour, the quad's vertices and the draw call -- and that state APPLIED, per the OpenGL ES 3.0 blend
    # equations for the factor names found, to sequences of the executed fragment shader's outputs (f32 destination).
    import re
    ts = open("/root/reference/src/renderers/WebGLRenderer.ts").read()
    ts = re.sub(r"/\*.*?\*/", "", ts, flags=re.S)
    ts = "\n".join(l for l in ts.split("\n") if not l.lstrip().startswith("//"))
    names = lambda call: [sorted(set(tuple(a.strip().replace("gl.", "") for a in m.split(","))
                                     for m in re.findall(r"gl\." + call + r"\(([^)]*)\)", ts)))]
    state = {"blendFuncSeparate": names("blendFuncSeparate")[0], "blendEquationSeparate": names("blendEquationSeparate")[0],
             "clearColor": names("clearColor")[0], "drawArraysInstanced": [a[:3] for a in names("drawArraysInstanced")[0]],
             "quad": [float(x) for x in re.search(r"triangleVertices\s*=\s*new Float32Array\(\[([^\]]*)\]\)", ts).group(1).split(",")]}
    assert all(len(v) == 1 for k, v in state.items() if k != "quad"), state      # one state for every draw
    srgb, drgb, sa, da = state["blendFuncSeparate"][0]
    factor = {"ONE": lambda S, D: F(1.0), "ZERO": lambda S, D: F(0.0), "ONE_MINUS_DST_ALPHA": lambda S, D: F(1.0) - D[3],
              "ONE_MINUS_SRC_ALPHA": lambda S, D: F(1.0) - S[3], "SRC_ALPHA": lambda S, D: S[3], "DST_ALPHA": lambda S, D: D[3]}
    assert state["blendEquationSeparate"][0] == ("FUNC_ADD", "FUNC_ADD")
    seqs = []
    with np.errstate(all="ignore"):
        for k in range(24):
            D = [F(x) for x in state["clearColor"][0]]
            items = []
            for j in range(3 + k % 6):
                r = 1.9 * np.sqrt(rng.random()) if j % 4 else 2.2
                a = 2.0 * np.pi * rng.random()
                vpos = (F(r * np.cos(a)), F(r * np.sin(a)))
                col = [F(x) for x in rng.random(4)]
                if k % 5 == 0:
                    col[3] = F(0.97)                    # nearly opaque layers: the destination saturates
                items.append([hexf(x) for x in vpos] + [hexf(x) for x in col])
                frag_ns["vPosition"] = frag_ns["vec2"](*vpos)
                frag_ns["vColor"] = frag_ns["vec4"](*col)
                try:
                    frag_ns["main"]()
                except G.Discard:
                    continue
                S = frag_ns["fragColor"].c
                fs = [factor[srgb](S, D)] * 3 + [factor[sa](S, D)]
                fd = [factor[drgb](S, D)] * 3 + [factor[da](S, D)]
                D = [S[c] * fs[c] + D[c] * fd[c] for c in range(4)]
            seqs.append({"fragments": items, "dst": [hexf(x) for x in D]})
    out["gl_state"] = {k: [list(t) for t in v] if k != "quad" else v for k, v in state.items()}
    out["blend_sequences"] = seqs

    # eval_sh_rgb: seeded half-packed coefficient textures, degrees 0..3, sample directions
    nsh = 24
    shs = (rng.random(nsh * 48).astype(np.float32) - 0.5) * 1.6
    sc2 = gh.Scene()
    sc2.bandsIndices = np.array([-1, 7, 15], dtype=np.int32)
    sc2.setData(np.array(gh.synth.synth_rows(nsh, 3)), shs)
    texs = [G.Texture([int(w) for w in t], 2048) for t in sc2.shs_rgb]
    sh_cases = []
    with np.errstate(all="ignore"):
        for k in range(nsh):
            d = rng.normal(size=3)
            d = [F(x) for x in d / np.linalg.norm(d)]
            for deg in range(4):
                rgb = vert_ns["eval_sh_rgb"](texs[0], texs[1], texs[2], k, deg, vert_ns["vec3"](*d))
                sh_cases.append({"index": k, "deg": deg, "dir": [hexf(x) for x in d], "rgb": vec_bits(rgb)})
    out["sh"] = {"count": nsh, "shs": shs.tobytes().hex(), "words": [np.asarray(t[:8 * nsh], dtype=np.uint32).tobytes().hex() for t in sc2.shs_rgb],
                 "cases": sh_cases}
    json.dump(out, open(os.path.join(HERE, "shader_golden.json"), "w"))
    print("wrote shader_golden.json:", len(out["cameras"]), "cameras x", n, "splats,", len(frags), "fragments,", len(sh_cases), "SH evaluations")


if __name__ == "__main__":
    main()
