"use strict";
// Scene container: API of src/core/Scene.ts.  setData() is the producer of the two buffers the render hot path
// consumes -- `positions` (f32 x 3N, read by the depth sort) and `data` (8 u32 per splat: position bits, six
// truncated halves of 4*Sigma, rgba8; the reference's RGBA32UI texture image) -- so its f64 arithmetic follows
// Scene.ts:126-177 operation for operation.  translate/rotate/scale/limitBox mutate the same buffers and fire
// "change", which makes the renderer re-upload (WebGLRenderer.ts:234-239).
const { EventDispatcher } = require("./EventDispatcher");
const { Matrix3 } = require("../math/Matrix3");
const { Quaternion } = require("../math/Quaternion");
const { Vector3 } = require("../math/Vector3");
const { packHalf2x16 } = require("../utils");

const ROW = 32;        // bytes per .splat row (Scene.ts:9)
const TEX_WIDTH = 2048; // texels per data-texture row, 2 texels per splat (Scene.ts:47)

class Scene extends EventDispatcher {
    constructor() {
        super();
        this._data = new Uint32Array(0);
        this._vertexCount = 0;
        this._width = TEX_WIDTH;
        this._height = 0;
        this._shHeight = 0;
        this._positions = new Float32Array(0);
        this._rotations = new Float32Array(0);
        this._scales = new Float32Array(0);
        this._shs = new Uint32Array(0);
        this._shs_rgb = [new Uint32Array(0), new Uint32Array(0), new Uint32Array(0)];
        this._g0bands = 0;
        this._bandsIndices = new Int32Array([-1, -1, -1]);
    }

    // 4*Sigma of splat i from its rotation/scale, packed as six truncated halves (Scene.ts:150-176)
    _packCovariance(i) {
        const r = this._rotations, s = this._scales;
        const rot = Matrix3.RotationFromQuaternion(new Quaternion(r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], -r[4 * i]));
        const M = Matrix3.Diagonal(new Vector3(s[3 * i], s[3 * i + 1], s[3 * i + 2])).multiply(rot).buffer;
        const d = this._data;
        d[8 * i + 4] = packHalf2x16(4 * (M[0] * M[0] + M[3] * M[3] + M[6] * M[6]), 4 * (M[0] * M[1] + M[3] * M[4] + M[6] * M[7]));
        d[8 * i + 5] = packHalf2x16(4 * (M[0] * M[2] + M[3] * M[5] + M[6] * M[8]), 4 * (M[1] * M[1] + M[4] * M[4] + M[7] * M[7]));
        d[8 * i + 6] = packHalf2x16(4 * (M[1] * M[2] + M[4] * M[5] + M[7] * M[8]), 4 * (M[2] * M[2] + M[5] * M[5] + M[8] * M[8]));
    }

    _writePosition(i) {
        const f = new Float32Array(this._data.buffer, this._data.byteOffset, this._data.length);
        f[8 * i] = this._positions[3 * i];
        f[8 * i + 1] = this._positions[3 * i + 1];
        f[8 * i + 2] = this._positions[3 * i + 2];
    }

    // data: Uint8Array of 32-byte rows [pos f32x3 | scale f32x3 | rgba u8x4 | rot u8x4 (w,x,y,z)].
    // shs (48 floats per SH-carrying splat) is packed into three half textures exactly like Scene.ts:108-124.
    setData(data, shs) {
        if (data.length % ROW) throw new Error("splat data length must be a multiple of " + ROW);
        const n = data.length / ROW;
        this._vertexCount = n;
        this._height = Math.ceil((2 * n) / this._width);
        this._data = new Uint32Array(this._width * this._height * 4);
        this._positions = new Float32Array(3 * n);
        this._rotations = new Float32Array(4 * n);
        this._scales = new Float32Array(3 * n);
        const bytes = data.byteOffset % 4 === 0 ? data : new Uint8Array(data);  // aligned view for the f32 reads
        const rowF = new Float32Array(bytes.buffer, bytes.byteOffset, n * 8);
        if (shs !== undefined) {
            const shCount = n - (this._bandsIndices[0] + 1);
            this._shHeight = Math.ceil((2 * shCount) / this._width);
            this._shs_rgb = [0, 1, 2].map(() => new Uint32Array(this._width * this._shHeight * 4));
            for (let i = 0; i < shCount; i++)
                for (let j = 0, src = i * 48; j < 8; j++, src += 6)
                    for (let c = 0; c < 3; c++) this._shs_rgb[c][8 * i + j] = packHalf2x16(shs[src + c], shs[src + 3 + c]);
        }
        const out8 = new Uint8Array(this._data.buffer);
        for (let i = 0; i < n; i++) {
            for (let k = 0; k < 3; k++) {
                this._positions[3 * i + k] = rowF[8 * i + k];
                this._scales[3 * i + k] = rowF[8 * i + 3 + k];
            }
            for (let k = 0; k < 4; k++) {
                this._rotations[4 * i + k] = (bytes[ROW * i + 28 + k] - 128) / 128;
                out8[4 * (8 * i + 7) + k] = bytes[ROW * i + 24 + k];
            }
            this._writePosition(i);
            this._packCovariance(i);
        }
        this.dispatchEvent({ type: "change" });
    }

    translate(t) {
        for (let i = 0; i < this._vertexCount; i++) {
            this._positions[3 * i] += t.x;
            this._positions[3 * i + 1] += t.y;
            this._positions[3 * i + 2] += t.z;
            this._writePosition(i);
        }
        this.dispatchEvent({ type: "change" });
    }

    rotate(rotation) {
        const R = Matrix3.RotationFromQuaternion(rotation).buffer;
        const p = this._positions, r = this._rotations;
        for (let i = 0; i < this._vertexCount; i++) {
            const x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            p[3 * i] = R[0] * x + R[1] * y + R[2] * z;
            p[3 * i + 1] = R[3] * x + R[4] * y + R[5] * z;
            p[3 * i + 2] = R[6] * x + R[7] * y + R[8] * z;
            this._writePosition(i);
            const q = rotation.multiply(new Quaternion(r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], r[4 * i]));
            r[4 * i + 1] = q.x; r[4 * i + 2] = q.y; r[4 * i + 3] = q.z; r[4 * i] = q.w;
            this._packCovariance(i);
        }
        this.dispatchEvent({ type: "change" });
    }

    scale(s) {
        const f = [s.x, s.y, s.z];
        for (let i = 0; i < this._vertexCount; i++) {
            for (let k = 0; k < 3; k++) {
                this._positions[3 * i + k] *= f[k];
                this._scales[3 * i + k] *= f[k];
            }
            this._writePosition(i);
            this._packCovariance(i);
        }
        this.dispatchEvent({ type: "change" });
    }

    limitBox(xMin, xMax, yMin, yMax, zMin, zMax) {
        if (xMin >= xMax) throw new Error("xMin (" + xMin + ") must be smaller than xMax (" + xMax + ")");
        if (yMin >= yMax) throw new Error("yMin (" + yMin + ") must be smaller than yMax (" + yMax + ")");
        if (zMin >= zMax) throw new Error("zMin (" + zMin + ") must be smaller than zMax (" + zMax + ")");
        const p = this._positions;
        let kept = 0;
        for (let i = 0; i < this._vertexCount; i++) {
            const x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            if (!(x >= xMin && x <= xMax && y >= yMin && y <= yMax && z >= zMin && z <= zMax)) continue;
            this._data.copyWithin(8 * kept, 8 * i, 8 * i + 8);
            this._positions.copyWithin(3 * kept, 3 * i, 3 * i + 3);
            this._rotations.copyWithin(4 * kept, 4 * i, 4 * i + 4);
            this._scales.copyWithin(3 * kept, 3 * i, 3 * i + 3);
            kept++;
        }
        this._height = Math.ceil((2 * kept) / this._width);
        this._vertexCount = kept;
        this._data = new Uint32Array(this._data.buffer, 0, this._width * this._height * 4);
        this._positions = new Float32Array(this._positions.buffer, 0, 3 * kept);
        this._rotations = new Float32Array(this._rotations.buffer, 0, 4 * kept);
        this._scales = new Float32Array(this._scales.buffer, 0, 3 * kept);
        this.dispatchEvent({ type: "change" });
    }

    // The 32-byte .splat rows of the current scene (what Scene.saveToFile downloads in a browser, Scene.ts:368-403).
    toSplatBytes() {
        const n = this._vertexCount;
        const out = new Uint8Array(n * ROW), outF = new Float32Array(out.buffer), src8 = new Uint8Array(this._data.buffer);
        for (let i = 0; i < n; i++) {
            for (let k = 0; k < 3; k++) {
                outF[8 * i + k] = this._positions[3 * i + k];
                outF[8 * i + 3 + k] = this._scales[3 * i + k];
            }
            for (let k = 0; k < 4; k++) {
                out[ROW * i + 24 + k] = src8[4 * (8 * i + 7) + k];
                out[ROW * i + 28 + k] = (this._rotations[4 * i + k] * 128 + 128) & 0xff;
            }
        }
        return out;
    }

    // Node replacement for the browser download: writes the rows with fs.
    saveToFile(name) { require("fs").writeFileSync(name, this.toSplatBytes()); }

    updateColor() {}
}

// plain accessors, as in Scene.ts:414-508
for (const k of ["data", "vertexCount", "width", "height", "positions", "rotations", "scales", "shs", "shs_rgb", "shHeight",
                 "g0bands", "bandsIndices"]) {
    Object.defineProperty(Scene.prototype, k, {
        get() { return this["_" + k]; },
        set(v) { this["_" + k] = v; },
        configurable: true,
    });
}
Scene.RowLength = ROW;
module.exports = { Scene };
