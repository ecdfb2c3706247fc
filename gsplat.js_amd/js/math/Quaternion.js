"use strict";
// Immutable quaternion (x, y, z, w), f64.  API of src/math/Quaternion.ts; FromEuler / multiply keep the
// reference's term order because camera poses feed the bit-exact sort path.
const { Vector3 } = require("./Vector3");

class Quaternion {
    constructor(x, y, z, w) {
        this.x = x === undefined ? 0 : x;
        this.y = y === undefined ? 0 : y;
        this.z = z === undefined ? 0 : z;
        this.w = w === undefined ? 1 : w;
    }
    equals(q) { return this.x === q.x && this.y === q.y && this.z === q.z && this.w === q.w; }
    normalize() {
        const l = Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z + this.w * this.w);
        return new Quaternion(this.x / l, this.y / l, this.z / l, this.w / l);
    }
    // Hamilton product this * q (src/math/Quaternion.ts:39-55)
    multiply(q) {
        const a = this;
        return new Quaternion(
            a.w * q.x + a.x * q.w + a.y * q.z - a.z * q.y,
            a.w * q.y - a.x * q.z + a.y * q.w + a.z * q.x,
            a.w * q.z + a.x * q.y - a.y * q.x + a.z * q.w,
            a.w * q.w - a.x * q.x - a.y * q.y - a.z * q.z);
    }
    flat() { return [this.x, this.y, this.z, this.w]; }
    clone() { return new Quaternion(this.x, this.y, this.z, this.w); }
    // src/math/Quaternion.ts:65-83
    static FromEuler(e) {
        const hx = e.x / 2, hy = e.y / 2, hz = e.z / 2;
        const cy = Math.cos(hy), sy = Math.sin(hy);
        const cp = Math.cos(hx), sp = Math.sin(hx);
        const cz = Math.cos(hz), sz = Math.sin(hz);
        return new Quaternion(
            cy * sp * cz + sy * cp * sz,
            sy * cp * cz - cy * sp * sz,
            cy * cp * sz - sy * sp * cz,
            cy * cp * cz + sy * sp * sz);
    }
    // src/math/Quaternion.ts:85-104
    toEuler() {
        const x = Math.atan2(2 * (this.w * this.x + this.y * this.z), 1 - 2 * (this.x * this.x + this.y * this.y));
        const sinp = 2 * (this.w * this.y - this.z * this.x);
        const y = Math.abs(sinp) >= 1 ? (Math.sign(sinp) * Math.PI) / 2 : Math.asin(sinp);
        const z = Math.atan2(2 * (this.w * this.z + this.x * this.y), 1 - 2 * (this.y * this.y + this.z * this.z));
        return new Vector3(x, y, z);
    }
    // src/math/Quaternion.ts:106-135 (row-major 3x3)
    static FromMatrix3(matrix) {
        const m = matrix.buffer;
        const trace = m[0] + m[4] + m[8];
        if (trace > 0) {
            const s = 0.5 / Math.sqrt(trace + 1.0);
            return new Quaternion((m[7] - m[5]) * s, (m[2] - m[6]) * s, (m[3] - m[1]) * s, 0.25 / s);
        }
        if (m[0] > m[4] && m[0] > m[8]) {
            const s = 2.0 * Math.sqrt(1.0 + m[0] - m[4] - m[8]);
            return new Quaternion(0.25 * s, (m[1] + m[3]) / s, (m[2] + m[6]) / s, (m[7] - m[5]) / s);
        }
        if (m[4] > m[8]) {
            const s = 2.0 * Math.sqrt(1.0 + m[4] - m[0] - m[8]);
            return new Quaternion((m[1] + m[3]) / s, 0.25 * s, (m[5] + m[7]) / s, (m[2] - m[6]) / s);
        }
        const s = 2.0 * Math.sqrt(1.0 + m[8] - m[0] - m[4]);
        return new Quaternion((m[2] + m[6]) / s, (m[5] + m[7]) / s, 0.25 * s, (m[3] - m[1]) / s);
    }
}
module.exports = { Quaternion };
