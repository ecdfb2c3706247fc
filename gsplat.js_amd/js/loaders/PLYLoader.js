"use strict";
// INRIA / 3DGS .ply reader for Node: src/loaders/PLYLoader.ts without the browser (fetch -> fs).
//   LoadAsync(file, scene, onProgress, format = "", useShs = false, quantized = false)
// format "" or "polycam" (y/z swap + 90 degree rotation about x, PLYLoader.ts:513-520).
// useShs = false: rows only (_ParsePLYBuffer, :389-538): any float/int properties, colour from f_dc_* as
//   (0.5 + SH_C0 * f_dc) * 255 or from red/green/blue, alpha = sigmoid(opacity) * 255, bytes written through a
//   Uint8ClampedArray (round half to even, clamp), rotation normalised to (w,x,y,z) bytes q*128+128.
// useShs = true: rows + 48 SH floats per splat (_ParseFullPLYBufferFast, :578-712), reproduced as written, including
//   its two oddities: the colour byte is 0.5 + SH_C0 * f_dc * 255 (precedence as in the source) and the slot of
//   f_rest_39 is filled from f_rest_38.  Scene.bandsIndices stays (-1,-1,-1): every splat is SH degree 3.
// quantized = true (with useShs): the codebook-compressed variant (_ParseQPLYBuffer, :893-1196): four vertex elements
//   vertex_0..vertex_3 (splats with 0, 1, 2, 3 SH bands), positions as half floats, every other property a one-byte
//   index into a 256-entry codebook (half floats, stored [256][codebooks] after the vertex data); yields the rows, 48 SH
//   floats for every splat of vertex_1..3 (unused tail zero) and Scene.bandsIndices = last index with 0 / 1 / 2 bands.
const fs = require("fs");
const { Scene } = require("../core/Scene");
const { Quaternion } = require("../math/Quaternion");
const { Vector3 } = require("../math/Vector3");

const SH_C0 = 0.28209479177387814;
const TYPE_SIZE = { double: 8, int: 4, uint: 4, float: 4, short: 2, ushort: 2, uchar: 1 };

class PLYLoader {
    static async LoadAsync(file, scene, onProgress, format, useShs, quantized) {
        const buf = await fs.promises.readFile(file);
        if (onProgress) onProgress(1, true);
        return PLYLoader.LoadFromBytes(new Uint8Array(buf.buffer, buf.byteOffset, buf.byteLength), scene, format, useShs, quantized);
    }
    static async LoadFromFileAsync(file, scene, onProgress, format, useShs, quantized) {
        return PLYLoader.LoadAsync(file, scene, onProgress, format, useShs, quantized);
    }
    static LoadFromBytes(bytes, scene, format, useShs, quantized) {
        if (bytes[0] !== 112 || bytes[1] !== 108 || bytes[2] !== 121 || bytes[3] !== 10) throw new Error("Invalid PLY file");
        const ab = bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength);
        if (useShs && quantized) {
            const parsed = PLYLoader._parseQuantized(ab);
            scene.bandsIndices = parsed[2];   // before setData, like PLYLoader.ts:85
            scene.setData(new Uint8Array(parsed[0]), new Float32Array(parsed[1]));
        } else if (useShs) {
            const parsed = PLYLoader._parseFull(PLYLoader._parseHeader(ab), ab);
            scene.setData(new Uint8Array(parsed[0]), new Float32Array(parsed[1]));
        } else {
            scene.setData(new Uint8Array(PLYLoader._parseRows(ab, format === undefined ? "" : format)));
        }
        return scene;
    }

    // PLYLoader.ts:541-575
    static _parseHeader(ab) {
        const text = Buffer.from(ab, 0, Math.min(ab.byteLength, 10240)).toString("utf8");
        const marker = "end_header\n";
        const end = text.indexOf(marker);
        if (end < 0) throw new Error("Unable to read .ply file header");
        const vertexCount = parseInt(/element vertex (\d+)\n/.exec(text)[1]);
        const properties = [];
        let rowOffset = 0;
        for (const line of text.slice(0, end).split("\n")) {
            if (!line.startsWith("property ")) continue;
            const parts = line.split(" ");
            if (!TYPE_SIZE[parts[1]]) throw new Error("Unsupported property type: " + parts[1]);
            properties.push({ name: parts[2], type: parts[1], offset: rowOffset });
            rowOffset += TYPE_SIZE[parts[1]];
        }
        return { properties: properties, size: end + marker.length, vertexCount: vertexCount, rowOffset: rowOffset };
    }

    static _writeRotation(rot, q) {
        const n = q.normalize();
        rot[0] = n.w * 128 + 128; rot[1] = n.x * 128 + 128; rot[2] = n.y * 128 + 128; rot[3] = n.z * 128 + 128;
    }

    // Rows only (what PLYLoader.ts:389-538 produces).  The header is compiled once into a list of field writers --
    // (byte offset, reader for the property's type, destination field, slot, value transform) for every property the
    // .splat row has a use for, in FILE order, which is what decides the outcome when two properties feed one byte
    // (red and f_dc_0, opacity and f_dc_3) -- and the vertex loop just runs that list.  Byte fields go through
    // Uint8ClampedArray stores (round half to even, clamp), floats through Float32Array stores.
    static _rowWriters(properties) {
        const ident = (v) => v, dcToByte = (v) => (0.5 + SH_C0 * v) * 255;
        const POS = 0, SCALE = 1, RGBA = 2, QUAT = 3;
        const families = [   // name pattern -> destination field, transform; the slot is the axis letter or trailing digit
            [/^[xyz]$/, POS, ident], [/^scale_[012]$/, SCALE, Math.exp], [/^rot_[0123]$/, QUAT, ident],
            [/^f_dc_[0123]$/, RGBA, dcToByte],
        ];
        const singles = { red: [RGBA, 0, ident], green: [RGBA, 1, ident], blue: [RGBA, 2, ident],
                          opacity: [RGBA, 3, (v) => (1 / (1 + Math.exp(-v))) * 255] };
        const readers = { float: DataView.prototype.getFloat32, int: DataView.prototype.getInt32 };
        const writers = [];
        for (const p of properties) {
            if (!readers[p.type]) throw new Error("Unsupported property type: " + p.type);
            let target = singles.hasOwnProperty(p.name) ? singles[p.name] : null;
            for (let k = 0; !target && k < families.length; k++) {
                if (!families[k][0].test(p.name)) continue;
                const last = p.name[p.name.length - 1];
                target = [families[k][1], last >= "x" ? last.charCodeAt(0) - 120 : Number(last), families[k][2]];
            }
            if (target) writers.push({ offset: p.offset, read: readers[p.type], field: target[0], slot: target[1], xf: target[2] });
        }
        return writers;
    }

    static _parseRows(ab, format) {
        if (format !== "" && format !== "polycam") throw new Error("Unsupported format: " + format);
        const h = PLYLoader._parseHeader(ab);
        const writers = PLYLoader._rowWriters(h.properties);
        const view = new DataView(ab, h.size);
        const out = new ArrayBuffer(Scene.RowLength * h.vertexCount);
        const polycam = format === "polycam" ? Quaternion.FromEuler(new Vector3(Math.PI / 2, 0, 0)) : null;
        const quat = new Float64Array(4);   // rot_0..rot_3 = (w, x, y, z)
        for (let i = 0; i < h.vertexCount; i++) {
            const base = i * Scene.RowLength;
            const position = new Float32Array(out, base, 3);
            const fields = [position, new Float32Array(out, base + 12, 3), new Uint8ClampedArray(out, base + 24, 4), quat];
            quat[0] = 255; quat[1] = quat[2] = quat[3] = 0;   // a file without rot_* gets these (:443)
            for (const w of writers) fields[w.field][w.slot] = w.xf(w.read.call(view, w.offset + i * h.rowOffset, true));
            let q = new Quaternion(quat[1], quat[2], quat[3], quat[0]);
            if (polycam) {   // y/z swap and a quarter turn about x (:513-520)
                const y = position[1];
                position[1] = -position[2];
                position[2] = y;
                q = polycam.multiply(q);
            }
            PLYLoader._writeRotation(new Uint8ClampedArray(out, base + 28, 4), q);
        }
        return out;
    }

    // PLYLoader.ts:578-712: rows + SH coefficients (k, channel) at 3k + channel; f_rest_j holds channel floor(j/15)
    static _parseFull(h, ab) {
        const view = new DataView(ab, h.size);
        const rows = new ArrayBuffer(Scene.RowLength * h.vertexCount);
        const shs = new ArrayBuffer(192 * h.vertexCount);
        const prop = {};
        for (const p of h.properties) prop[p.name] = p;
        const stride = h.properties[h.properties.length - 1].offset + 4;
        // source slot of each of the 45 higher-band floats, as listed in the reference (the 30th entry repeats f_rest_38)
        const restOrder = [];
        for (let k = 0; k < 15; k++) for (let c = 0; c < 3; c++) restOrder.push(k + 15 * c);
        restOrder[29] = 38;
        const f = (name, i) => view.getFloat32(prop[name].offset + i * stride, true);
        for (let i = 0; i < h.vertexCount; i++) {
            const position = new Float32Array(rows, i * Scene.RowLength, 3);
            const scale = new Float32Array(rows, i * Scene.RowLength + 12, 3);
            const rgba = new Uint8ClampedArray(rows, i * Scene.RowLength + 24, 4);
            const rot = new Uint8ClampedArray(rows, i * Scene.RowLength + 28, 4);
            const sh = new Float32Array(shs, i * 192, 48);
            position[0] = f("x", i); position[1] = f("y", i); position[2] = f("z", i);
            scale[0] = Math.exp(f("scale_0", i)); scale[1] = Math.exp(f("scale_1", i)); scale[2] = Math.exp(f("scale_2", i));
            rgba[0] = 0.5 + SH_C0 * f("f_dc_0", i) * 255;
            rgba[1] = 0.5 + SH_C0 * f("f_dc_1", i) * 255;
            rgba[2] = 0.5 + SH_C0 * f("f_dc_2", i) * 255;
            rgba[3] = (1 / (1 + Math.exp(-f("opacity", i)))) * 255;
            PLYLoader._writeRotation(rot, new Quaternion(f("rot_1", i), f("rot_2", i), f("rot_3", i), f("rot_0", i)));
            sh[0] = f("f_dc_0", i); sh[1] = f("f_dc_1", i); sh[2] = f("f_dc_2", i);
            for (let j = 0; j < 45; j++) sh[3 + j] = f("f_rest_" + restOrder[j], i);
        }
        return [rows, shs];
    }

    // utils.ts:52-71 decodeFloat16 for one value (the reference stores the result in a Float32Array: exact)
    static _halfToFloat(bits) {
        const exponent = (bits & 0x7c00) >> 10, fraction = bits & 0x03ff;
        const sign = (bits & 0x8000) ? -1 : 1;
        if (exponent === 0) return sign * 6.103515625e-5 * (fraction / 0x400);
        if (exponent === 0x1f) return fraction ? NaN : sign * Infinity;
        return sign * Math.pow(2, exponent - 15) * (1 + fraction / 0x400);
    }

    // PLYLoader.ts:893-1196
    static _parseQuantized(ab) {
        const text = Buffer.from(ab, 0, Math.min(ab.byteLength, 10240)).toString("utf8");
        const marker = "end_header\n";
        const end = text.indexOf(marker);
        if (end < 0) throw new Error("Unable to read .ply file header");
        const cbStart = text.indexOf("element codebook_centers 256\n");
        const counts = [], starts = [];
        const re = /element vertex_(\d+) (\d+)/g;
        for (let m = re.exec(text); m; m = re.exec(text)) { counts.push(parseInt(m[2])); starts.push(m.index); }
        if (counts.length !== 4 || cbStart < 0) throw new Error("not a quantized PLY: expected vertex_0..vertex_3 and codebook_centers");
        const extents = [[0, starts[1]], [starts[1], starts[2]], [starts[2], starts[3]], [starts[3], cbStart]];
        const props = [], rowSize = [];
        let dataBytes = 0, total = 0;
        for (let e = 0; e < 4; e++) {
            const byName = {};
            let off = 0;
            for (const line of text.slice(extents[e][0], extents[e][1]).split("\n")) {
                if (!line.startsWith("property ")) continue;
                const parts = line.split(" ");
                if (!TYPE_SIZE[parts[1]]) throw new Error("Unsupported property type: " + parts[1]);
                byName[parts[2]] = off;
                off += TYPE_SIZE[parts[1]];
            }
            props.push(byName); rowSize.push(off);
            dataBytes += counts[e] * off; total += counts[e];
        }
        // codebooks: [256][nb] half floats right after the vertex data
        const cbNames = [];
        for (const line of text.slice(cbStart, end).split("\n")) if (line.startsWith("property ")) cbNames.push(line.split(" ")[2]);
        const nb = cbNames.length;
        const cbView = new DataView(ab, dataBytes + end + marker.length, nb * 2 * 256);
        const cb = {};
        for (let j = 0; j < nb; j++) {
            const t = new Float32Array(256);
            for (let i = 0; i < 256; i++) t[i] = PLYLoader._halfToFloat(cbView.getUint16(i * nb * 2 + j * 2, true));
            cb[cbNames[j]] = t;
        }
        const view = new DataView(ab, end + marker.length, dataBytes);
        const rows = new ArrayBuffer(Scene.RowLength * total);
        const shs = new ArrayBuffer(192 * (counts[1] + counts[2] + counts[3]));
        const strideLut = [3, 8, 15];
        const rest0 = props[1]["f_rest_0"];   // the reference takes this offset from vertex_1 for every element (:1053)
        let writeOff = 0, readOff = 0, shOff = 0;
        for (let e = 0; e < 4; e++) {
            const pr = props[e], rs = rowSize[e];
            const shStride = e > 0 ? strideLut[e - 1] : 0;
            let nRest = 0;
            for (const name in pr) if (name.startsWith("f_rest")) nRest++;
            const u8 = (name, v) => view.getUint8(readOff + pr[name] + v * rs);
            for (let v = 0; v < counts[e]; v++) {
                const position = new Float32Array(rows, writeOff + v * Scene.RowLength, 3);
                const scale = new Float32Array(rows, writeOff + v * Scene.RowLength + 12, 3);
                const rgba = new Uint8ClampedArray(rows, writeOff + v * Scene.RowLength + 24, 4);
                const rot = new Uint8ClampedArray(rows, writeOff + v * Scene.RowLength + 28, 4);
                for (let k = 0; k < 3; k++) position[k] = PLYLoader._halfToFloat(view.getUint16(readOff + pr["xyz"[k]] + v * rs, true));
                for (let k = 0; k < 3; k++) scale[k] = Math.exp(cb["scaling"][u8("scale_" + k, v)]);
                PLYLoader._writeRotation(rot, new Quaternion(cb["rotation_im"][u8("rot_1", v)], cb["rotation_im"][u8("rot_2", v)],
                                                             cb["rotation_im"][u8("rot_3", v)], cb["rotation_re"][u8("rot_0", v)]));
                for (let k = 0; k < 3; k++) rgba[k] = (0.5 + SH_C0 * cb["features_dc"][u8("f_dc_" + k, v)]) * 255;
                rgba[3] = (1 / (1 + Math.exp(-cb["opacity"][u8("opacity", v)]))) * 255;
                if (e > 0) {
                    const sh = new Float32Array(shs, shOff + v * 192, 48);
                    for (let k = 0; k < 3; k++) sh[k] = cb["features_dc"][u8("f_dc_" + k, v)];
                    // output slot 3 + m holds coefficient floor(m/3) of channel m % 3; the file is channel-major (:1146-1153)
                    for (let m = 0; m < nRest; m++) {
                        const coef = Math.floor(m / 3);
                        const idx = view.getUint8(readOff + rest0 + coef + shStride * (m % 3) + v * rs);
                        sh[3 + m] = cb["features_rest_" + coef][idx];
                    }
                }
            }
            writeOff += counts[e] * Scene.RowLength;
            readOff += counts[e] * rs;
            if (e > 0) shOff += counts[e] * 192;
        }
        const ind0 = counts[0] - 1, ind1 = ind0 + counts[1], ind2 = ind1 + counts[2];
        return [rows, shs, new Int32Array([ind0, ind1, ind2])];
    }
}
module.exports = { PLYLoader };
