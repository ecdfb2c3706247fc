#!/usr/bin/env node
// Generates tests/golden/host_golden.json by RUNNING the reference's own host classes -- src/math/{Vector3,Quaternion,
// Matrix3,Matrix4}.ts, src/core/{EventDispatcher,Object3D,Scene}.ts, src/cameras/Camera.ts, src/utils.ts -- under Node:
// the files are read from /root/reference at generation time, their TypeScript annotations are stripped by the small
// rewriter below (imports / exports, member declarations, parameter / return / variable types, `as`, `!`, generics), and
// the resulting classes are evaluated in one function scope.  Nothing of the reference's source is written to this
// repository: the fixture holds inputs and outputs only (bytes as hex, f64 as their bit patterns).
//   D1  Camera.update (Camera.ts:32-56,81-92), Matrix4.multiply (Matrix4.ts:32-53), Matrix3.RotationFromQuaternion /
//       multiply (Matrix3.ts:33-80), Quaternion.multiply / normalize (Quaternion.ts:34-57)
//   D2  Scene.setData incl. the SH packing (Scene.ts:58-180), translate / rotate / scale / limitBox (Scene.ts:182-366)
// Run in the build container:  node tests/golden/make_golden_host.js
"use strict";
const fs = require("fs");
const path = require("path");
const REF = "/root/reference/src/";

const TYPES = "(?:number|Number|string|boolean|any|void|null|Event|KeyboardEvent|MouseEvent|WheelEvent|TouchEvent|HTMLElement|Matrix3|Matrix4|Quaternion|Vector3|Camera|Scene|Float32Array|Uint8Array|Uint32Array|Int32Array|Int16Array)";
const TYPE = TYPES + "(?:\\[\\])?(?:\\s*\\|\\s*" + TYPES + "(?:\\[\\])?)*";
function strip(name) {
  let s = fs.readFileSync(REF + name, "utf8");
  const lines = s.split("\n").filter((ln) => {
    if (/^\s*import\s/.test(ln) || /^\s*export\s/.test(ln)) return false;
    // member declarations without an initialiser: `private _data: Uint32Array;`, `setData: (data: ...) => void;`
    if (/^ {4}(?:(?:public|private|protected|readonly)\s+)*[A-Za-z_]\w*[?!]?\s*:\s.*;\s*$/.test(ln) && !/\s=\s/.test(ln)) return false;
    // overload signatures: `add(v: Vector3): Vector3;`
    if (/^ {4}[A-Za-z_]\w*\(.*\)\s*:\s*[\w\[\]| ]+;\s*$/.test(ln)) return false;
    return true;
  });
  s = lines.join("\n");
  s = s.replace(/^( {4}[A-Za-z_]\w*)\s*:\s*\([^)]*\)\s*=>\s*\w+\s*=\s*/gm, "$1 = ");       // `attach: (c: Camera) => void = () => {};`
  s = s.replace(/:\s*\{\s*\[\w+:\s*\w+\]:\s*\w+\s*\}/g, "");                              // `const keys: { [key: string]: boolean } = {}`
  s = s.replace(/new (Map|Set)<.*>\(/g, "new $1(");                                   // generics
  s = s.replace(/\)!(?=[.)])/g, ")");                                                      // non-null assertions
  s = s.replace(/([A-Za-z_]\w*)\s*:\s*\(\w+\s*:\s*\w+\)\s*=>\s*\w+(?=\s*[,)])/g, "$1");    // function-typed parameters
  s = s.replace(/\s+as\s+[A-Z]\w*/g, "");                                             // `{ type: "change" } as Event`
  s = s.replace(/:\s*\[Uint32Array, Uint32Array, Uint32Array\]/g, "");               // the one tuple type
  s = s.replace(new RegExp("\\)\\s*:\\s*" + TYPE + "\\s*(?=\\{|=>)", "g"), ") ");     // return types
  s = s.replace(new RegExp("([A-Za-z_]\\w*)\\??\\s*:\\s*" + TYPE + "(?=\\s*[,)=;])", "g"), "$1");   // parameters, variables
  return s;
}

const order = ["utils.ts", "math/Vector3.ts", "math/Quaternion.ts", "math/Matrix3.ts", "math/Matrix4.ts",
               "core/EventDispatcher.ts", "core/Object3D.ts", "cameras/Camera.ts", "core/Scene.ts", "controls/OrbitControls.ts"];
const body = order.map(strip).join("\n") +
  "\nreturn { Vector3, Quaternion, Matrix3, Matrix4, Camera, Scene, OrbitControls, packHalf2x16 };";
let R;
try {
  const quiet = { log() {} };   // (Scene.setData and Camera.setFromData print to the console)
  const fakeWindow = { addEventListener() {}, removeEventListener() {} };   // (OrbitControls registers its key handlers there)
  R = new Function("console", "window", body)(quiet, fakeWindow);
} catch (e) {
  console.error("the stripped sources do not evaluate:", e.message);
  process.exit(1);
}

// ---- helpers: bit patterns ----
const f64 = new Float64Array(1), b8 = new Uint8Array(f64.buffer);
const hex64 = (x) => { f64[0] = x; return Array.from(b8).reverse().map((b) => b.toString(16).padStart(2, "0")).join(""); };
const hexBytes = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString("hex");
let seed = 20240601;
const rnd = () => { seed = (Math.imul(seed, 1664525) + 1013904223) >>> 0; return seed / 4294967296; };
const rndS = (a) => (rnd() * 2 - 1) * a;

const out = { generator: "tests/golden/make_golden_host.js",
              source: "src/{utils,math/*,core/EventDispatcher,core/Object3D,core/Scene,cameras/Camera}.ts evaluated under node " + process.version };

// ---- D1: matrices, quaternions, cameras ----
const quats = [];
for (let k = 0; k < 24; k++) {
  let q = [rndS(1), rndS(1), rndS(1), rndS(1)];
  if (k === 0) q = [0, 0, 0, 1];
  if (k === 1) q = [0.5, -0.5, 0.5, 0.5];
  quats.push(q);
}
out.quaternions = quats.map((q) => {
  const Q = new R.Quaternion(q[0], q[1], q[2], q[3]), N = Q.normalize();
  const other = new R.Quaternion(q[3], -q[1], q[0], q[2]);
  return { q: q.map(hex64), normalized: N.flat().map(hex64), times_other: Q.multiply(other).flat().map(hex64),
           other: other.flat().map(hex64),
           rotation_of_normalized: R.Matrix3.RotationFromQuaternion(N).buffer.map(hex64),
           rotation_raw: R.Matrix3.RotationFromQuaternion(Q).buffer.map(hex64) };
});
out.matrix3_products = [];
out.matrix4_products = [];
for (let k = 0; k < 8; k++) {
  const a = Array.from({ length: 9 }, () => rndS(3)), b = Array.from({ length: 9 }, () => rndS(3));
  out.matrix3_products.push({ a: a.map(hex64), b: b.map(hex64), a_multiply_b: new R.Matrix3(...a).multiply(new R.Matrix3(...b)).buffer.map(hex64) });
  const c = Array.from({ length: 16 }, () => rndS(3)), d = Array.from({ length: 16 }, () => rndS(3));
  out.matrix4_products.push({ a: c.map(hex64), b: d.map(hex64), a_multiply_b: new R.Matrix4(...c).multiply(new R.Matrix4(...d)).buffer.map(hex64) });
}
out.cameras = [];
const sizes = [[1920, 1080, 1132, 1132], [640, 480, 1132, 1132], [3840, 2160, 2264, 2264], [1000, 700, 900.5, 1100.25]];
for (let k = 0; k < 12; k++) {
  const q = new R.Quaternion(rndS(1), rndS(1), rndS(1), rndS(1)).normalize();
  const p = [rndS(9), rndS(9), rndS(9)];
  const [w, h, fx, fy] = sizes[k % sizes.length];
  const near = k % 5 === 4 ? 0.2 : 0.01, far = k % 5 === 4 ? 250 : 1000;
  const cam = new R.Camera(new R.Vector3(p[0], p[1], p[2]), q, fx, fy, near, far);
  cam.update(w, h);
  out.cameras.push({ position: p.map(hex64), rotation: q.flat().map(hex64), width: w, height: h, fx: hex64(fx), fy: hex64(fy),
                     near: hex64(near), far: hex64(far), projectionMatrix: cam.projectionMatrix.buffer.map(hex64),
                     viewMatrix: cam.viewMatrix.buffer.map(hex64), viewProj: cam.viewProj.buffer.map(hex64) });
}

// ---- D2: Scene.setData, the SH packing, the transforms ----
const NROWS = 160;
const rows = new Uint8Array(NROWS * 32), rf = new Float32Array(rows.buffer);
for (let i = 0; i < NROWS; i++) {
  for (let k = 0; k < 3; k++) rf[8 * i + k] = rndS(4);
  for (let k = 0; k < 3; k++) rf[8 * i + 3 + k] = Math.exp(Math.log(0.004) + rnd() * (Math.log(0.06) - Math.log(0.004)));
  for (let k = 24; k < 32; k++) rows[32 * i + k] = (rnd() * 256) >>> 0;
}
// edge rows: identity rotation, zero scale, huge / tiny scales (half overflow, the shift-count quirk), alpha 0, big positions
rows.set([255, 128, 128, 128], 28);
rf[8 * 1 + 3] = 0; rf[8 * 1 + 4] = 0; rf[8 * 1 + 5] = 0;
rf[8 * 2 + 3] = 200; rf[8 * 2 + 4] = 1e-9; rf[8 * 2 + 5] = 3;
rows[32 * 3 + 27] = 0;
rf[8 * 4 + 3] = 1e-20; rf[8 * 4 + 4] = 1e-12; rf[8 * 4 + 5] = 1e-7;
rf[8 * 5 + 0] = 1e6; rf[8 * 5 + 1] = -3.5e4; rf[8 * 5 + 2] = 65504;
rows.set([0, 0, 0, 0], 32 * 6 + 28);
rows.set([255, 255, 255, 255], 32 * 7 + 28);
const snapshot = (sc) => ({ vertexCount: sc.vertexCount, width: sc.width, height: sc.height, data_length: sc.data.length,
  data: hexBytes(new Uint32Array(sc.data.buffer, sc.data.byteOffset, 8 * sc.vertexCount)),
  data_tail_is_zero: Array.from(sc.data.subarray(8 * sc.vertexCount)).every((x) => x === 0),
  positions: hexBytes(sc.positions), rotations: hexBytes(sc.rotations), scales: hexBytes(sc.scales) });
{
  const sc = new R.Scene();
  let events = 0;
  sc.addEventListener("change", () => { events++; });
  sc.setData(rows.slice());
  out.scene = { rows: hexBytes(rows), after_setData: snapshot(sc) };
  const t = [0.25, -0.5, 1.0];
  sc.translate(new R.Vector3(t[0], t[1], t[2]));
  out.scene.translate = { t: t.map(hex64), after: snapshot(sc) };
  const q = new R.Quaternion(0.3, -0.2, 0.1, 0.9).normalize();
  sc.rotate(q);
  out.scene.rotate = { q: q.flat().map(hex64), after: snapshot(sc) };
  const s = [1.5, 0.75, 1.25];
  sc.scale(new R.Vector3(s[0], s[1], s[2]));
  out.scene.scale = { s: s.map(hex64), after: snapshot(sc) };
  const box = [-3, 3, -2.5, 4, -3.5, 3];
  sc.limitBox(...box);
  out.scene.limitBox = { box: box.map(hex64), after: snapshot(sc) };
  out.scene.change_events = events;
}
{
  // SH: splats [first, n) carry 48 coefficients each (Scene.ts:83-124); bandsIndices[0] = first - 1
  const first = 40, nsh = NROWS - first;
  const shs = new Float32Array(nsh * 48);
  for (let i = 0; i < shs.length; i++) shs[i] = rndS(0.6);
  shs[0] = 0; shs[1] = 1e-9; shs[2] = 70000; shs[3] = -0.3;
  const sc = new R.Scene();
  sc.bandsIndices = new Int32Array([first - 1, 80, 120]);
  sc.setData(rows.slice(), shs);
  out.scene_sh = { first: first, shs: hexBytes(shs), shHeight: sc.shHeight,
                   shs_rgb: sc.shs_rgb.map((t) => hexBytes(new Uint32Array(t.buffer, t.byteOffset, 8 * nsh))),
                   tails_are_zero: sc.shs_rgb.every((t) => Array.from(t.subarray(8 * nsh)).every((x) => x === 0)),
                   texture_words: sc.shs_rgb.map((t) => t.length) };
}
{
  // OrbitControls (controls/OrbitControls.ts:20-307) on an element that only records its listeners: the pose the constructor
  // leaves (it ends with update()), then setCameraTarget and three damped updates -- camera.position / camera.rotation each time
  const el = { listeners: {}, addEventListener(k, f) { this.listeners[k] = f; }, removeEventListener() {} };
  const pose = (cam) => ({ position: cam.position.flat().map(hex64), rotation: cam.rotation.flat().map(hex64) });
  out.orbit = [];
  for (let k = 0; k < 10; k++) {
    const alpha = k === 0 ? 0.5 : rndS(3), beta = k === 0 ? 0.5 : rndS(1.4), radius = k === 0 ? 5 : 0.5 + rnd() * 9;
    const target = k < 2 ? [0, 0, 0] : [rndS(2), rndS(2), rndS(2)];
    const cam = new R.Camera();
    const oc = new R.OrbitControls(cam, el, alpha, beta, radius, false, new R.Vector3(target[0], target[1], target[2]));
    const rec = { alpha: hex64(alpha), beta: hex64(beta), radius: hex64(radius), target: target.map(hex64), dampening: hex64(oc.dampening),
                  after_constructor: pose(cam), steps: [] };
    oc.update();
    rec.steps.push(pose(cam));
    const nt = [rndS(1.5), rndS(1.5), rndS(1.5)];
    oc.setCameraTarget(new R.Vector3(nt[0], nt[1], nt[2]));
    rec.new_target = nt.map(hex64);
    for (let j = 0; j < 3; j++) { oc.update(); rec.steps.push(pose(cam)); }
    out.orbit.push(rec);
    oc.dispose();
  }
}
fs.writeFileSync(path.join(__dirname, "host_golden.json"), JSON.stringify(out));
console.log("wrote host_golden.json:", out.cameras.length, "cameras,", NROWS, "rows");
