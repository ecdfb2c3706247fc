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
// The quantized-PLY variant with codebooks (_ParseQPLYBuffer, :893-1196) is not implemented.
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
        if (quantized) throw new Error("quantized PLY (codebooks) is not supported by this loader");
        const ab = bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength);
        if (useShs) {
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

    // PLYLoader.ts:389-538
    static _parseRows(ab, format) {
        if (format !== "" && format !== "polycam") throw new Error("Unsupported format: " + format);
        const h = PLYLoader._parseHeader(ab);
        const view = new DataView(ab, h.size);
        const out = new ArrayBuffer(Scene.RowLength * h.vertexCount);
        const qPolycam = Quaternion.FromEuler(new Vector3(Math.PI / 2, 0, 0));
        for (let i = 0; i < h.vertexCount; i++) {
            const position = new Float32Array(out, i * Scene.RowLength, 3);
            const scale = new Float32Array(out, i * Scene.RowLength + 12, 3);
            const rgba = new Uint8ClampedArray(out, i * Scene.RowLength + 24, 4);
            const rot = new Uint8ClampedArray(out, i * Scene.RowLength + 28, 4);
            let r0 = 255, r1 = 0, r2 = 0, r3 = 0;
            for (const p of h.properties) {
                let v;
                if (p.type === "float") v = view.getFloat32(p.offset + i * h.rowOffset, true);
                else if (p.type === "int") v = view.getInt32(p.offset + i * h.rowOffset, true);
                else throw new Error("Unsupported property type: " + p.type);
                switch (p.name) {
                    case "x": position[0] = v; break;
                    case "y": position[1] = v; break;
                    case "z": position[2] = v; break;
                    case "scale_0": scale[0] = Math.exp(v); break;
                    case "scale_1": scale[1] = Math.exp(v); break;
                    case "scale_2": scale[2] = Math.exp(v); break;
                    case "red": rgba[0] = v; break;
                    case "green": rgba[1] = v; break;
                    case "blue": rgba[2] = v; break;
                    case "f_dc_0": rgba[0] = (0.5 + SH_C0 * v) * 255; break;
                    case "f_dc_1": rgba[1] = (0.5 + SH_C0 * v) * 255; break;
                    case "f_dc_2": rgba[2] = (0.5 + SH_C0 * v) * 255; break;
                    case "f_dc_3": rgba[3] = (0.5 + SH_C0 * v) * 255; break;
                    case "opacity": rgba[3] = (1 / (1 + Math.exp(-v))) * 255; break;
                    case "rot_0": r0 = v; break;
                    case "rot_1": r1 = v; break;
                    case "rot_2": r2 = v; break;
                    case "rot_3": r3 = v; break;
                    default: break;
                }
            }
            let q = new Quaternion(r1, r2, r3, r0);
            if (format === "polycam") {
                const t = position[1];
                position[1] = -position[2];
                position[2] = t;
                q = qPolycam.multiply(q);
            }
            PLYLoader._writeRotation(rot, q);
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
}
module.exports = { PLYLoader };
