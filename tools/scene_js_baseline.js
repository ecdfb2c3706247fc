#!/usr/bin/env node
"use strict";
// What Scene.setData and Scene.rotate cost on the host, in JavaScript: the loops of src/core/Scene.ts:126-177 and :196-255
// as the package's Scene runs them (gsplat.js_amd/js/core/Scene.js: the same per-splat arithmetic in f64, Matrix3 / Quaternion
// objects per splat like the reference), beside which bench.py prints the device kernels' times (gsr_set_scene_rows,
// gsr_scene_rotate: k_scene.hip).   node tools/scene_js_baseline.js <n>   -> one JSON line
const path = require("path");
const G = require(path.join(__dirname, "..", "gsplat.js_amd", "js"));
const n = +(process.argv[2] || 1000000);
const rows = new Uint8Array(n * 32), f = new Float32Array(rows.buffer);
let s = 12345;
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
for (let i = 0; i < n; i++) {
    for (let k = 0; k < 3; k++) f[8 * i + k] = (rnd() * 2 - 1) * 4;
    for (let k = 0; k < 3; k++) f[8 * i + 3 + k] = 0.004 + rnd() * 0.05;
    for (let k = 24; k < 32; k++) rows[32 * i + k] = (rnd() * 256) >>> 0;
}
const now = () => Number(process.hrtime.bigint()) * 1e-6;
const scene = new G.Scene();
let t0 = now();
scene.setData(rows);
const tSet = now() - t0;
t0 = now();
scene.rotate(new G.Quaternion(0.3, -0.2, 0.1, 0.9).normalize());
const tRot = now() - t0;
t0 = now();
scene.translate(new G.Vector3(0.25, -0.5, 1));
const tTr = now() - t0;
console.log(JSON.stringify({ host: "node " + process.version, n: n, setData_ms: tSet, rotate_ms: tRot, translate_ms: tTr,
                             setData_ns_per_splat: tSet * 1e6 / n, rotate_ns_per_splat: tRot * 1e6 / n }));
