#!/usr/bin/env node
"use strict";
// Frames per second through the JavaScript host -- the reference's own calling convention, renderer.render(scene, camera)
// (WebGLRenderer.ts:241-296), one synchronous frame after the other -- and with three renderers used round-robin
// (renderAsync / sync, what bench.py's default measures through ctypes).  The scene is a .splat file (32-byte rows);
// scripts/final_lines.sh writes the synthetic C3 scene with the Python generator first.
//   node tools/bench_node.js <scene.splat> <W> <H> <fx> [frames] [warmup]
// prints one JSON line.
const path = require("path");
const G = require(path.join(__dirname, "..", "gsplat.js_amd", "js"));
const [file, W, H, fx, framesArg, warmArg] = process.argv.slice(2);
const frames = framesArg ? +framesArg : 240, warm = warmArg ? +warmArg : 20;

function orbitCamera(k) {
    const cam = new G.Camera(undefined, undefined, +fx, +fx);
    G.OrbitControls.applyPose(cam, (2 * Math.PI * (k % 120)) / 120, 0.3, 8, new G.Vector3(0, 0, 0));
    return cam;
}
const now = () => Number(process.hrtime.bigint()) * 1e-6;   // ms

const scene = new G.Scene();
G.Loader.LoadSync(file, scene);
const cams = [];
for (let k = 0; k < 120; k++) cams.push(orbitCamera(k));

// (1) the reference's contract: render() returns when the frame is done
const r = new G.WebGLRenderer({ width: +W, height: +H }, []);
for (let k = 0; k < warm; k++) r.render(scene, cams[k % 120]);
let t0 = now();
for (let k = 0; k < frames; k++) r.render(scene, cams[(warm + k) % 120]);
const syncMs = (now() - t0) / frames;
// with the pixels read back as RGBA8 every frame (what a consumer without a display does)
t0 = now();
let sink = 0;
const nread = Math.min(frames, 120);
for (let k = 0; k < nread; k++) { r.render(scene, cams[k % 120]); sink += r.readPixels()[1]; }
const readMs = (now() - t0) / nread;
const pixels = new Uint8Array(+W * +H * 4);   // reused, like the destination of gl.readPixels
r.render(scene, cams[0]); r.readPixels(pixels);
t0 = now();
for (let k = 0; k < nread; k++) { r.render(scene, cams[k % 120]); sink += r.readPixels(pixels)[1]; }
const readReuseMs = (now() - t0) / nread;
r.dispose();

// (2) three renderers, frames in flight
const rs = [0, 1, 2].map(() => new G.WebGLRenderer({ width: +W, height: +H, throughput: true }, []));
for (let k = 0; k < warm; k++) rs[k % 3].renderAsync(scene, cams[k % 120]);
rs.forEach((x) => x.sync());
t0 = now();
for (let k = 0; k < frames; k++) rs[k % 3].renderAsync(scene, cams[(warm + k) % 120]);
rs.forEach((x) => x.sync());
const asyncMs = (now() - t0) / frames;
const info = rs[0].deviceInfo ? rs[0].deviceInfo() : null;
rs.forEach((x) => x.dispose());

console.log(JSON.stringify({
    host: "node " + process.version + ", gsplat.js_amd/js (N-API addon over libgsplat_hip.so)",
    scene: path.basename(file), splats: scene.vertexCount, width: +W, height: +H, frames: frames, warmup: warm,
    render_sync: { frames_per_sec: 1000 / syncMs, ms_per_frame: syncMs, note: "renderer.render(scene, camera), one frame at a time" },
    render_sync_with_readPixels: { frames_per_sec: 1000 / readMs, ms_per_frame: readMs, note: "fresh Uint8Array per frame" },
    render_sync_with_readPixels_reused_array: { frames_per_sec: 1000 / readReuseMs, ms_per_frame: readReuseMs },
    render_async_3_renderers: { frames_per_sec: 1000 / asyncMs, ms_per_frame: asyncMs },
    device: info, sink: sink & 1,
}));
