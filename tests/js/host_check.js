#!/usr/bin/env node
// Driver used by tests/test_js_host.py: exercises the JavaScript host through its public API and dumps
// binary results for the Python side to compare with the oracle.
"use strict";
const fs = require("fs");
const path = require("path");
const G = require(path.join(__dirname, "..", "..", "gsplat.js_amd", "js"));

const [, , mode, ...a] = process.argv;
const writeBin = (file, typed) => fs.writeFileSync(file, Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength));

function orbitCamera(k, frames, fx) {
    const cam = new G.Camera(undefined, undefined, fx, fx);
    G.OrbitControls.applyPose(cam, (2 * Math.PI * k) / frames, 0.3, 8, new G.Vector3(0, 0, 0));
    return cam;
}

if (mode === "pack") {                     // pack <splat> <outprefix> <W> <H> <fx> <pose...>
    const [file, out, W, H, fx, ...poses] = a;
    const scene = new G.Scene();
    let events = 0;
    scene.addEventListener("change", () => events++);
    G.Loader.LoadSync(file, scene);
    writeBin(out + ".data.bin", scene.data);
    writeBin(out + ".pos.bin", scene.positions);
    const cams = poses.map((k) => {
        const cam = orbitCamera(+k, 120, +fx);
        cam.update(+W, +H);
        return { pose: +k, position: cam.position.flat(), rotation: cam.rotation.flat(), view: cam.viewMatrix.buffer,
                 proj: cam.projectionMatrix.buffer, viewProj: cam.viewProj.buffer };
    });
    // transforms: translate, rotate, scale, limitBox on a copy
    const s2 = new G.Scene();
    G.Loader.LoadSync(file, s2);
    s2.translate(new G.Vector3(0.25, -0.5, 1));
    const q = G.Quaternion.FromEuler(new G.Vector3(0.1, 0.2, 0.3));
    s2.rotate(q);
    s2.scale(new G.Vector3(1.5, 1.5, 1.5));
    s2.limitBox(-4, 4, -4, 4, -4, 4);
    writeBin(out + ".xf.splat", s2.toSplatBytes());
    writeBin(out + ".xf.data.bin", s2.data);
    writeBin(out + ".xf.pos.bin", s2.positions);
    writeBin(out + ".xf.rot.bin", s2.rotations);
    writeBin(out + ".xf.scl.bin", s2.scales);
    fs.writeFileSync(out + ".json", JSON.stringify({ events, vertexCount: scene.vertexCount, width: scene.width, height: scene.height,
                                                     dataLength: scene.data.length, cams, xfCount: s2.vertexCount, q: q.flat() }));
} else if (mode === "packsh") {             // packsh <splat> <shs.f32> <outprefix> <b0> <b1> <b2>
    const [file, shfile, out, b0, b1, b2] = a;
    const scene = new G.Scene();
    scene.bandsIndices = new Int32Array([+b0, +b1, +b2]);
    const raw = fs.readFileSync(shfile);
    const shs = new Float32Array(raw.buffer.slice(raw.byteOffset, raw.byteOffset + raw.byteLength));
    const rows = fs.readFileSync(file);
    const bytes = new Uint8Array(rows.length);
    bytes.set(rows);
    scene.setData(bytes, shs);
    for (let c = 0; c < 3; c++) writeBin(out + ".sh" + c + ".bin", scene.shs_rgb[c]);
    fs.writeFileSync(out + ".json", JSON.stringify({ shHeight: scene.shHeight, vertexCount: scene.vertexCount }));
} else if (mode === "render") {            // render <splat> <outprefix> <W> <H> <fx> <pose> [eps]
    const [file, out, W, H, fx, pose, eps] = a;
    const scene = new G.Scene();
    G.Loader.LoadSync(file, scene);
    const r = new G.WebGLRenderer({ width: +W, height: +H, earlyOutEps: eps ? +eps : 0, timing: true }, []);   // no passes: steady state
    const cam = orbitCamera(+pose, 120, +fx);
    r.render(scene, cam);
    writeBin(out + ".depthIndex.bin", r.lastDepthIndex());
    writeBin(out + ".rgba32f.bin", r.readPixelsFloat());
    writeBin(out + ".rgba8.bin", r.readPixels());
    // scene mutation must trigger a re-upload through the "change" event (WebGLRenderer.ts:234-239)
    scene.translate(new G.Vector3(0.5, 0, 0));
    r.render(scene, cam);
    writeBin(out + ".moved.depthIndex.bin", r.lastDepthIndex());
    writeBin(out + ".moved.pos.bin", scene.positions);
    // wasm drop-in
    const di = new Uint32Array(scene.vertexCount), keys = new Uint32Array(scene.vertexCount);
    G.sortHost(new Float32Array(cam.viewProj.buffer), scene.vertexCount, scene.positions, keys, di);
    writeBin(out + ".sortHost.depthIndex.bin", di);
    fs.writeFileSync(out + ".json", JSON.stringify({ stats: r.stats(), device: r.deviceInfo(), viewProj: cam.viewProj.buffer }));
    r.dispose();
} else if (mode === "inflight") {          // inflight <splat> <outprefix> <W> <H> <fx>: three throughput renderers used round-robin
    const [file, out, W, H, fx] = a;
    const scene = new G.Scene();
    G.Loader.LoadSync(file, scene);
    const rs = [0, 1, 2].map(() => new G.WebGLRenderer({ width: +W, height: +H, throughput: true }, []));
    const ref = new G.WebGLRenderer({ width: +W, height: +H }, []);
    let same = true, frames = 0;
    const pending = [null, null, null];
    const check = (slot) => {
        const r = rs[slot];
        r.sync();
        ref.render(scene, orbitCamera(pending[slot], 120, +fx));
        const a1 = r.lastDepthIndex(), a2 = ref.lastDepthIndex();
        for (let i = 0; same && i < a1.length; i++) same = a1[i] === a2[i];
        const p1 = r.readPixels(), p2 = ref.readPixels();
        for (let i = 0; same && i < p1.length; i++) same = Math.abs(p1[i] - p2[i]) <= 1;
        frames++;
    };
    for (let k = 0; k < 12; k++) {
        const slot = k % 3;
        if (pending[slot] !== null) check(slot);
        rs[slot].renderAsync(scene, orbitCamera(7 * k, 120, +fx));
        pending[slot] = 7 * k;
    }
    for (let slot = 0; slot < 3; slot++) check(slot);
    fs.writeFileSync(out + ".json", JSON.stringify({ same, frames }));
    rs.forEach((r) => r.dispose()); ref.dispose();
} else if (mode === "golden") {
    // golden <host_golden.json>: this package's classes on the inputs of the fixture that tests/golden/make_golden_host.js
    // produced by running the reference's own TypeScript; prints the outputs that differ (none expected) and how many were compared
    const g = JSON.parse(fs.readFileSync(a[0], "utf8"));
    const f64 = new Float64Array(1), b8 = new Uint8Array(f64.buffer);
    const un = (h) => { for (let i = 0; i < 8; i++) b8[7 - i] = parseInt(h.substr(2 * i, 2), 16); return f64[0]; };
    const hex64 = (x) => { f64[0] = x; return Array.from(b8).reverse().map((b) => b.toString(16).padStart(2, "0")).join(""); };
    const hexBytes = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString("hex");
    const bytes = (h) => { const b = Buffer.from(h, "hex"), u = new Uint8Array(b.length); u.set(b); return u; };
    const bad = [];
    let compared = 0;
    const same = (what, got, want) => { compared++; if (JSON.stringify(got) !== JSON.stringify(want)) bad.push(what); };
    g.quaternions.forEach((e, k) => {
        const q = e.q.map(un), Q = new G.Quaternion(q[0], q[1], q[2], q[3]), N = Q.normalize(), o = e.other.map(un);
        same("quaternion.normalize " + k, N.flat().map(hex64), e.normalized);
        same("quaternion.multiply " + k, Q.multiply(new G.Quaternion(o[0], o[1], o[2], o[3])).flat().map(hex64), e.times_other);
        same("RotationFromQuaternion(normalized) " + k, G.Matrix3.RotationFromQuaternion(N).buffer.map(hex64), e.rotation_of_normalized);
        same("RotationFromQuaternion(raw) " + k, G.Matrix3.RotationFromQuaternion(Q).buffer.map(hex64), e.rotation_raw);
    });
    g.matrix3_products.forEach((e, k) => same("Matrix3.multiply " + k, new G.Matrix3(...e.a.map(un)).multiply(new G.Matrix3(...e.b.map(un))).buffer.map(hex64), e.a_multiply_b));
    g.matrix4_products.forEach((e, k) => same("Matrix4.multiply " + k, new G.Matrix4(...e.a.map(un)).multiply(new G.Matrix4(...e.b.map(un))).buffer.map(hex64), e.a_multiply_b));
    g.cameras.forEach((e, k) => {
        const p = e.position.map(un), q = e.rotation.map(un);
        const cam = new G.Camera(new G.Vector3(p[0], p[1], p[2]), new G.Quaternion(q[0], q[1], q[2], q[3]), un(e.fx), un(e.fy), un(e.near), un(e.far));
        cam.update(e.width, e.height);
        same("Camera.projectionMatrix " + k, cam.projectionMatrix.buffer.map(hex64), e.projectionMatrix);
        same("Camera.viewMatrix " + k, cam.viewMatrix.buffer.map(hex64), e.viewMatrix);
        same("Camera.viewProj " + k, cam.viewProj.buffer.map(hex64), e.viewProj);
    });
    const snap = (what, sc, w) => {
        same(what + " vertexCount", sc.vertexCount, w.vertexCount);
        same(what + " texture", [sc.width, sc.height, sc.data.length], [w.width, w.height, w.data_length]);
        same(what + " data", hexBytes(new Uint32Array(sc.data.buffer, sc.data.byteOffset, 8 * sc.vertexCount)), w.data);
        same(what + " positions", hexBytes(sc.positions), w.positions);
        same(what + " rotations", hexBytes(sc.rotations), w.rotations);
        same(what + " scales", hexBytes(sc.scales), w.scales);
    };
    {
        const sc = new G.Scene();
        let events = 0;
        sc.addEventListener("change", () => { events++; });
        sc.setData(bytes(g.scene.rows));
        snap("setData", sc, g.scene.after_setData);
        same("setData tail", Array.from(sc.data.subarray(8 * sc.vertexCount)).every((x) => x === 0), g.scene.after_setData.data_tail_is_zero);
        const t = g.scene.translate.t.map(un);
        sc.translate(new G.Vector3(t[0], t[1], t[2]));
        snap("translate", sc, g.scene.translate.after);
        const q = g.scene.rotate.q.map(un);
        sc.rotate(new G.Quaternion(q[0], q[1], q[2], q[3]));
        snap("rotate", sc, g.scene.rotate.after);
        const s = g.scene.scale.s.map(un);
        sc.scale(new G.Vector3(s[0], s[1], s[2]));
        snap("scale", sc, g.scene.scale.after);
        sc.limitBox(...g.scene.limitBox.box.map(un));
        snap("limitBox", sc, g.scene.limitBox.after);
        same("change events", events, g.scene.change_events);
    }
    {
        const e = g.scene_sh, sb = bytes(e.shs), shs = new Float32Array(sb.buffer, 0, sb.length / 4), nsh = shs.length / 48;
        const sc = new G.Scene();
        sc.bandsIndices = new Int32Array([e.first - 1, 80, 120]);
        sc.setData(bytes(g.scene.rows), shs);
        same("sh height", sc.shHeight, e.shHeight);
        same("sh texture words", sc.shs_rgb.map((t) => t.length), e.texture_words);
        for (let c = 0; c < 3; c++) same("sh channel " + c, hexBytes(new Uint32Array(sc.shs_rgb[c].buffer, sc.shs_rgb[c].byteOffset, 8 * nsh)), e.shs_rgb[c]);
    }
    (g.orbit || []).forEach((e, k) => {
        // OrbitControls (controls/OrbitControls.ts:20-307 executed): the pose the constructor leaves, then update / setCameraTarget / updates
        const t = e.target.map(un), nt = e.new_target.map(un);
        const cam = new G.Camera();
        const oc = new G.OrbitControls(cam, null, un(e.alpha), un(e.beta), un(e.radius), false, new G.Vector3(t[0], t[1], t[2]));
        const pose = () => ({ position: cam.position.flat().map(hex64), rotation: cam.rotation.flat().map(hex64) });
        same("OrbitControls dampening " + k, hex64(oc.dampening), e.dampening);
        same("OrbitControls constructor " + k, pose(), e.after_constructor);
        oc.update();
        same("OrbitControls update " + k, pose(), e.steps[0]);
        oc.setCameraTarget(new G.Vector3(nt[0], nt[1], nt[2]));
        for (let j = 1; j < 4; j++) { oc.update(); same("OrbitControls step " + j + " of " + k, pose(), e.steps[j]); }
    });
    console.log(JSON.stringify({ compared, mismatches: bad }));
} else if (mode === "group") {
    // group <splat> <outprefix> <W> <H> <fx> <rank> <world> <idfile>: renderer.render(scene, camera) with the framebuffer
    // all-gather inside the library (RCCL).  Rank 0 writes the communicator id to <idfile>, the others wait for it: the id
    // travels by whatever the host has.  world 1 = the self test the one-GPU box can run; on a multi-GPU node start one
    // process per rank with HIP_VISIBLE_DEVICES=<rank> (or { device: rank }).
    const [file, out, W, H, fx, rank, world, idfile] = a;
    const scene = new G.Scene();
    G.Loader.LoadSync(file, scene);
    let id;
    if (+rank === 0) {
        id = G.WebGLRenderer.createGroupId();
        fs.writeFileSync(idfile + ".tmp", Buffer.from(id));
        fs.renameSync(idfile + ".tmp", idfile);
    } else {
        const t0 = Date.now();
        while (!fs.existsSync(idfile)) { if (Date.now() - t0 > 60000) throw new Error("no communicator id after 60 s"); }
        id = new Uint8Array(fs.readFileSync(idfile));
    }
    const r = new G.WebGLRenderer({ width: +W, height: +H, device: +world > 1 ? +rank : 0 }, []);
    r.joinGroup({ id: id, rank: +rank, world: +world, edges: G.WebGLRenderer.bandEdges(+W, +world) });
    const ref = +rank === 0 ? new G.WebGLRenderer({ width: +W, height: +H }, []) : null;
    let worst = 0;
    for (const pose of [3, 47, 91]) {
        const cam = orbitCamera(pose, 120, +fx);
        r.render(scene, cam);
        const got = r.readPixels();
        if (ref) {
            ref.render(scene, cam);
            const want = ref.readPixels();
            for (let i = 0; i < want.length; i++) worst = Math.max(worst, Math.abs(got[i] - want[i]));
        }
    }
    // a second renderer of this rank shares the group (one communicator and exchange stream per rank): frames in flight with
    // renderAsync, alternating renderers, each read back before its renderer is used again
    const r2 = new G.WebGLRenderer({ width: +W, height: +H, device: +world > 1 ? +rank : 0 }, []);
    r2.shareGroup(r);
    let worstShared = 0;
    const pair = [r, r2], posesAsync = [12, 55, 70, 101];
    const inflight = [null, null];
    const settle = (slot) => {
        const got = pair[slot].readPixels();
        if (ref) {
            ref.render(scene, orbitCamera(inflight[slot], 120, +fx));
            const want = ref.readPixels();
            for (let i = 0; i < want.length; i++) worstShared = Math.max(worstShared, Math.abs(got[i] - want[i]));
        }
        inflight[slot] = null;
    };
    posesAsync.forEach((pose, k) => {
        const slot = k % 2;
        if (inflight[slot] !== null) settle(slot);
        pair[slot].renderAsync(scene, orbitCamera(pose, 120, +fx));
        inflight[slot] = pose;
    });
    settle(0); settle(1);
    if (ref) fs.writeFileSync(out + ".json", JSON.stringify({ worst: worst, worstShared: worstShared, world: +world, group: r.group(), group2: r2.group() }));
    r2.leaveGroup();
    r2.dispose();
    r.leaveGroup();
    r.dispose();
    if (ref) ref.dispose();
} else if (mode === "renderfade") {       // renderfade <splat> <outprefix> <W> <H> <fx> <pose> <frames>: default passes = [FadeInPass]
    const [file, out, W, H, fx, pose, frames] = a;
    const scene = new G.Scene();
    G.Loader.LoadSync(file, scene);
    const r = new G.WebGLRenderer({ width: +W, height: +H });
    const cam = orbitCamera(+pose, 120, +fx);
    for (let k = 0; k < +frames; k++) r.render(scene, cam);
    writeBin(out + ".rgba32f.bin", r.readPixelsFloat());
    r.dispose();
} else if (mode === "devscene") {         // devscene <splat> <outprefix>: device-side setData + transforms vs the JS Scene
    const [file, out] = a;
    const rows = new Uint8Array(fs.readFileSync(file));
    const scene = new G.Scene();
    scene.setData(rows);
    const r = new G.WebGLRenderer({ width: 320, height: 240 }, []);
    r.setSceneRows(rows);
    const q = G.Quaternion.FromEuler(new G.Vector3(0.2, -0.4, 0.6));
    const t = new G.Vector3(0.5, 0.25, -1), sc = new G.Vector3(1.25, 0.8, 1.1);
    scene.translate(t); r.sceneTranslate(t);
    scene.rotate(q); r.sceneRotate(q);
    scene.scale(sc); r.sceneScale(sc);
    scene.limitBox(-3, 3, -3, 3, -3, 3); r.sceneLimitBox(-3, 3, -3, 3, -3, 3);
    const dev = r.readSceneData();
    let same = dev.vertexCount === scene.vertexCount;
    for (let i = 0; same && i < dev.vertexCount * 8; i++) same = dev.data[i] === scene.data[i];
    for (let i = 0; same && i < dev.vertexCount * 3; i++) same = Object.is(dev.positions[i], scene.positions[i]);
    const cam = orbitCamera(20, 120, 400);
    r.renderDeviceScene(cam);
    const a1 = r.readPixels();
    r.render(scene, cam);
    const a2 = r.readPixels();
    let samePixels = a1.length === a2.length;
    for (let i = 0; samePixels && i < a1.length; i++) samePixels = a1[i] === a2[i];
    fs.writeFileSync(out + ".json", JSON.stringify({ same, samePixels, n: dev.vertexCount }));
    r.dispose();
} else if (mode === "ply") {              // ply <file.ply> <outprefix>
    const [file, out] = a;
    const bytes = new Uint8Array(fs.readFileSync(file));
    const dump = (tag, fmt, useShs) => {
        const scene = new G.Scene();
        G.PLYLoader.LoadFromBytes(bytes, scene, fmt, useShs);
        writeBin(out + "." + tag + ".splat", scene.toSplatBytes());
        writeBin(out + "." + tag + ".data.bin", scene.data);
        if (useShs) for (let c = 0; c < 3; c++) writeBin(out + "." + tag + ".sh" + c + ".bin", scene.shs_rgb[c]);
        return scene.vertexCount;
    };
    const n = dump("plain", "", false);
    dump("polycam", "polycam", false);
    dump("full", "", true);
    let refused = false;
    try { G.PLYLoader.LoadFromBytes(bytes, new G.Scene(), "", true, true); } catch (e) { refused = true; }
    let badMagic = false;
    try { G.PLYLoader.LoadFromBytes(new Uint8Array(64), new G.Scene()); } catch (e) { badMagic = /Invalid PLY/.test(e.message); }
    fs.writeFileSync(out + ".json", JSON.stringify({ n, refused, badMagic }));
} else if (mode === "qply") {             // qply <file.ply> <outprefix>: the codebook-quantized variant
    const [file, out] = a;
    const bytes = new Uint8Array(fs.readFileSync(file));
    const ab = bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength);
    const parsed = G.PLYLoader._parseQuantized(ab);
    writeBin(out + ".rows.bin", new Uint8Array(parsed[0]));
    writeBin(out + ".shs.bin", new Float32Array(parsed[1]));
    const scene = new G.Scene();
    G.PLYLoader.LoadFromBytes(bytes, scene, "", true, true);
    for (let c = 0; c < 3; c++) writeBin(out + ".sh" + c + ".bin", scene.shs_rgb[c]);
    writeBin(out + ".splat", scene.toSplatBytes());
    fs.writeFileSync(out + ".json", JSON.stringify({ n: scene.vertexCount, bands: Array.from(scene.bandsIndices), parsedBands: Array.from(parsed[2]),
                                                     shHeight: scene.shHeight }));
} else if (mode === "plygolden") {        // plygolden <inria.ply> <quantized.ply>: what this package's parsers return, as hex (tests/test_host_golden.py)
    const ab = (f) => { const b = fs.readFileSync(f); return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength); };
    const hex = (buf) => Buffer.from(buf).toString("hex");
    const inria = ab(a[0]), q = ab(a[1]);
    const full = G.PLYLoader._parseFull(G.PLYLoader._parseHeader(inria), inria);
    const quant = G.PLYLoader._parseQuantized(q);
    console.log(JSON.stringify({ plain: hex(G.PLYLoader._parseRows(inria, "")), polycam: hex(G.PLYLoader._parseRows(inria, "polycam")),
                                 full_rows: hex(full[0]), full_shs: hex(full[1]),
                                 q_rows: hex(quant[0]), q_shs: hex(quant[1]), q_bands: Array.from(quant[2]) }));
} else if (mode === "nodevice") {
    try {
        new G.HIPRenderer({ width: 64, height: 64 });
        console.log("CREATED");
    } catch (e) {
        console.log("THROWN:" + e.message);
    }
} else if (mode === "api") {
    console.log(JSON.stringify(Object.keys(G).sort()));
} else {
    console.error("unknown mode");
    process.exit(2);
}
