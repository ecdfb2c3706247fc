#!/usr/bin/env node
// Generates tests/golden/half_golden.json by RUNNING the reference's own
// floatToHalf / packHalf2x16 (src/utils.ts:13-48): the function text is read from
// /root/reference at generation time, its TypeScript type annotations are stripped,
// and it is evaluated under Node.  Nothing of the reference's source is written to
// this repository: the fixture holds inputs (f64 bit patterns) and outputs only.
// Run in the build container:  node tests/golden/make_golden_half.js
"use strict";
const fs = require("fs");
const path = require("path");
const src = fs.readFileSync("/root/reference/src/utils.ts", "utf8");
const begin = src.indexOf("const _floatView");
const end = src.indexOf("var pow");
if (begin < 0 || end < 0) throw new Error("utils.ts layout changed");
let body = src.slice(begin, end);
body = body.replace(/:\s*(Float32Array|Int32Array|number)/g, "");
const mod = new Function(body + "\nreturn { floatToHalf: floatToHalf, packHalf2x16: packHalf2x16 };")();

const f64 = new Float64Array(1), u8 = new Uint8Array(f64.buffer);
const hex = (x) => { f64[0] = x; return Array.from(u8).reverse().map((b) => b.toString(16).padStart(2, "0")).join(""); };

const picks = [0, -0, 1, -1, 0.5, 0.3, -0.3, 2, 4, 1024, 32767, 32768, 32769, 65504, 65519, 65520, 65536, 1e5, 1e10, 3.4e38, 1e39,
  6.103515625e-5, 6.1e-5, 6.0e-5, 5.96e-8, 1e-7, 1e-8, 1e-10, 1e-15, 1e-20, 1e-30, 1e-38, 1e-39, 1e-45, 1.17549435e-38,
  0.1, 0.2, 0.7, 0.999, 1.001, 3.14159265, 2.71828, 0.0004, 0.0144, 0.000064, 123.456, 999.9, 2047.9, 2048.1, 4095.5,
  NaN, Infinity, -Infinity, -65504, -1e-7, -1e-15, -32768, 0.3000000119, 4 * 0.06 * 0.06, 4 * 0.004 * 0.004];
// plus a seeded sweep over every f32 exponent
let s = 12345;
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
const f32 = new Float32Array(1), i32 = new Int32Array(f32.buffer);
for (let e = 0; e < 256; e++) for (let k = 0; k < 6; k++) {
  i32[0] = ((rnd() < 0.5 ? 1 : 0) << 31) | (e << 23) | ((rnd() * 0x800000) >>> 0);
  picks.push(f32[0]);
}
const out = { generator: "tests/golden/make_golden_half.js", source: "src/utils.ts floatToHalf/packHalf2x16 run under node " + process.version,
  inputs_f64_hex: picks.map(hex), half: picks.map((x) => mod.floatToHalf(x)), pack_pairs: [] };
for (let i = 0; i + 1 < picks.length; i += 2) out.pack_pairs.push(mod.packHalf2x16(picks[i], picks[i + 1]));
fs.writeFileSync(path.join(__dirname, "half_golden.json"), JSON.stringify(out));
console.log("wrote", picks.length, "values");
