#!/usr/bin/env node
// Runs the reference's own PLY parsers -- PLYLoader.ts: _ParsePLYBuffer (:389-540), _parsePLYHeader + _ParseFullPLYBufferFast
// (:541-712, the pair LoadAsync calls with useShs) and _ParseQPLYBuffer (:893-1197) -- on the files named on the command line and
// prints what they return.  The static
// methods are cut out of the reference's file at generation time (brace matching from their signatures), their TypeScript
// annotations are erased (variable / parameter / return types, local `type` aliases, the one `?.`), and they are evaluated
// as plain functions together with the classes they use (utils, Vector3, Quaternion, Matrix3, Scene.RowLength).  Nothing
// of the reference's source is written anywhere: stdout is JSON with the outputs as hex.
//   node tests/golden/make_golden_ply.js <inria.ply> <quantized.ply>        (driver: tests/golden/make_golden_ply.py)
"use strict";
const fs = require("fs");
const REF = "/root/reference/src/";

function methodText(src, name, params) {
  const at = src.search(new RegExp("private static " + name + "\\("));
  if (at < 0) throw new Error("no method " + name);
  const sig = src.slice(at).match(/\)\s*:\s*[^{;]+\{/);           // `): ReturnType {`
  const open = at + sig.index + sig[0].length - 1;
  let depth = 0, i = open;
  for (; i < src.length; i++) {
    if (src[i] === "{") depth++;
    else if (src[i] === "}") { depth--; if (depth === 0) break; }
  }
  let body = src.slice(open, i + 1);
  body = body.replace(/\btype\s+\w+\s*=\s*\{[^}]*\};?/g, "");                                        // local type aliases
  body = body.replace(/((?:const|let|var)\s+[A-Za-z_]\w*)\s*:\s*[^=;]+?(?=\s*=[^=>])/g, "$1");       // `let x: T = ...`
  body = body.replace(/((?:const|let|var)\s+[A-Za-z_]\w*)\s*:\s*[^=;]+;/g, "$1;");                   // `let x: T;`
  body = body.replace(/([(,]\s*[A-Za-z_]\w*)\s*:\s*(?:number|string|boolean|any)(?=\s*[,)])/g, "$1"); // typed arrow parameters
  body = body.replace(/\(([^()]*:[^()]*)\)\s*:\s*\w+(\s*=>)/g, "($1)$2");                              // arrow return types: `(p: T, i) : number =>`
  body = body.replace(/\(([^()]*:[^()]*)\)(\s*=>)/g, (m, params, arrow) =>                           // `(acc: Record<string, T>, item: T) =>`
    "(" + params.replace(/<[^<>]*>/g, "").split(",").map((p) => p.replace(/\s*:.*$/, "")).join(",") + ")" + arrow);
  body = body.replace(/\)\?\.forEach\(/g, ").forEach(");                                            // Node 12 has no optional chaining
  body = body.replace(/\s+as\s+[A-Z]\w*(?:\[\])?/g, "");
  body = body.replace(/\)!(?=[\[.)])/g, ")");                                                       // non-null assertions
  return "function " + name + "(" + (params || "inputBuffer, format") + ") " + body;
}

const TYPES = "(?:number|Number|string|boolean|any|void|Event|Matrix3|Matrix4|Quaternion|Vector3|Float32Array|Uint8Array|Uint32Array|Int32Array|Int16Array)";
const TYPE = TYPES + "(?:\\[\\])?(?:\\s*\\|\\s*" + TYPES + "(?:\\[\\])?)*";
function stripClass(name) {          // (the rewriter of make_golden_host.js, for the small classes the parsers use)
  let s = fs.readFileSync(REF + name, "utf8");
  s = s.split("\n").filter((ln) => {
    if (/^\s*import\s/.test(ln) || /^\s*export\s/.test(ln)) return false;
    if (/^ {4}(?:(?:public|private|protected|readonly)\s+)*[A-Za-z_]\w*[?!]?\s*:\s.*;\s*$/.test(ln) && !/\s=\s/.test(ln)) return false;
    if (/^ {4}[A-Za-z_]\w*\(.*\)\s*:\s*[\w\[\]| ]+;\s*$/.test(ln)) return false;
    return true;
  }).join("\n");
  s = s.replace(new RegExp("\\)\\s*:\\s*" + TYPE + "\\s*(?=\\{|=>)", "g"), ") ");
  s = s.replace(new RegExp("([A-Za-z_]\\w*)\\??\\s*:\\s*" + TYPE + "(?=\\s*[,)=;])", "g"), "$1");
  return s;
}

const ply = fs.readFileSync(REF + "loaders/PLYLoader.ts", "utf8");
const shc0 = ply.match(/const\s+SH_C0\s*=\s*([0-9.]+);/)[1];
const body = ["utils.ts", "math/Vector3.ts", "math/Quaternion.ts", "math/Matrix3.ts"].map(stripClass).join("\n") +
  "\nconst SH_C0 = " + shc0 + ";\nconst Scene = { RowLength: " + fs.readFileSync(REF + "core/Scene.ts", "utf8").match(/static RowLength\s*=\s*([^;]+);/)[1] + " };\n" +
  ["_ParsePLYBuffer", "_parsePLYHeader", "_ParseQPLYBuffer"].map((n) => methodText(ply, n)).join("\n") + "\n" +
  methodText(ply, "_ParseFullPLYBufferFast", "header, inputBuffer, onProgress") +
  "\nreturn { _ParsePLYBuffer, _parsePLYHeader, _ParseFullPLYBufferFast, _ParseQPLYBuffer };";
let R;
try {
  R = new Function("console", "performance", body)({ log() {} }, { now: () => 0 });
} catch (e) {
  console.error("the erased sources do not evaluate:", e.message);
  if (process.env.GOLDEN_DEBUG) fs.writeFileSync(process.env.GOLDEN_DEBUG, "(function(console, performance){" + body + "})");
  process.exit(1);
}
const ab = (f) => { const b = fs.readFileSync(f); return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength); };
const hex = (buf) => Buffer.from(buf).toString("hex");
const inria = ab(process.argv[2]), q = ab(process.argv[3]);
const full = R._ParseFullPLYBufferFast(R._parsePLYHeader(inria, ""), inria);    // (what LoadAsync calls with useShs, PLYLoader.ts:85-86)
const quant = R._ParseQPLYBuffer(q, "");
process.stdout.write(JSON.stringify({
  generator: "tests/golden/make_golden_ply.py + make_golden_ply.js",
  source: "loaders/PLYLoader.ts: _ParsePLYBuffer, _parsePLYHeader + _ParseFullPLYBufferFast, _ParseQPLYBuffer evaluated under node " + process.version,
  plain: hex(R._ParsePLYBuffer(inria, "")), polycam: hex(R._ParsePLYBuffer(inria, "polycam")),
  full_rows: hex(full[0]), full_shs: hex(full[1]),
  q_rows: hex(quant[0]), q_shs: hex(quant[1]), q_bands: Array.from(quant[2]),
}));
