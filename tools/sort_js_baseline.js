"use strict";
// CPU baseline stand-in for the reference's wasm sort (SURVEY.md 8(d)): the four passes of wasm/wasm.cpp:8-52 written
// in plain JavaScript with typed arrays (Math.fround keeps every intermediate in f32 like the C++), timed on a
// positions file.  No emcc in the image, so the wasm blob itself cannot be rebuilt and the prebuilt one is not run;
// V8 running this loop is what a browser without wasm would do, and a fair upper bound on the worker's time.
//   node tools/sort_js_baseline.js <positions.f32> <vp2> <vp6> <vp10> [calls]
// prints {"n":..,"ms_median":..,"calls":..,"checksum":..}; checksum = FNV-1a of depthIndex for cross-checking.
const fs = require("fs");
const [file, a2, a6, a10, callsArg] = process.argv.slice(2);
const buf = fs.readFileSync(file);
const pos = new Float32Array(buf.buffer, buf.byteOffset, buf.byteLength >> 2);
const n = (pos.length / 3) | 0;
const vp2 = Math.fround(+a2), vp6 = Math.fround(+a6), vp10 = Math.fround(+a10);
const calls = callsArg ? +callsArg : 10;
const depthBuffer = new Uint32Array(n), depthIndex = new Uint32Array(n);
const RANGE = 65536;
const counts = new Uint32Array(RANGE + 1), starts = new Uint32Array(RANGE + 1);
const f = Math.fround;

function sort() {
    let minDepth = 0x7fffffff, maxDepth = -0x80000000;
    for (let i = 0; i < n; i++) {       // wasm.cpp:14-31
        const d = (f(f(f(f(vp2 * pos[3 * i]) + f(vp6 * pos[3 * i + 1])) + f(vp10 * pos[3 * i + 2])) * 4096)) | 0;
        depthBuffer[i] = d;
        if (d > maxDepth) maxDepth = d;
        if (d < minDepth) minDepth = d;
    }
    counts.fill(0);
    if (maxDepth !== minDepth) {        // wasm.cpp:33-40
        const depthInv = f(RANGE / f(maxDepth - minDepth));
        for (let i = 0; i < n; i++) {
            let q = f(f((depthBuffer[i] - minDepth) >>> 0) * depthInv) >>> 0;
            if (q > RANGE) q = RANGE;
            depthBuffer[i] = q;
            counts[q]++;
        }
    } else {
        depthBuffer.fill(0);
        counts[0] = n;
    }
    starts[0] = 0;                      // wasm.cpp:42-46 (+ the one extra bucket of the defined max-bucket semantics)
    for (let i = 1; i <= RANGE; i++) starts[i] = starts[i - 1] + counts[i - 1];
    for (let i = 0; i < n; i++) depthIndex[starts[depthBuffer[i]]++] = i;   // wasm.cpp:48-51
}

const times = [];
for (let c = 0; c < calls; c++) {
    const t0 = process.hrtime.bigint();
    sort();
    times.push(Number(process.hrtime.bigint() - t0) / 1e6);
}
times.sort((x, y) => x - y);
let h = 0xcbf29ce484222325n;
const bytes = new Uint8Array(depthIndex.buffer);
for (let i = 0; i < bytes.length; i++) { h ^= BigInt(bytes[i]); h = (h * 0x100000001b3n) & 0xffffffffffffffffn; }
console.log(JSON.stringify({ n: n, ms_median: times[times.length >> 1], calls: calls, checksum: h.toString(16) }));
