"use strict";
// Row-major 3x3, f64, immutable buffer.  API of src/math/Matrix3.ts.  The sums keep the reference's
// left-to-right order: Scene.setData's covariance bits depend on it.
class Matrix3 {
    constructor(n11, n12, n13, n21, n22, n23, n31, n32, n33) {
        const d = (v, dflt) => (v === undefined ? dflt : v);
        this.buffer = [d(n11, 1), d(n12, 0), d(n13, 0), d(n21, 0), d(n22, 1), d(n23, 0), d(n31, 0), d(n32, 0), d(n33, 1)];
    }
    equals(m) {
        if (this.buffer === m.buffer) return true;
        if (this.buffer.length !== m.buffer.length) return false;
        return this.buffer.every((v, i) => v === m.buffer[i]);
    }
    // this * m (src/math/Matrix3.ts:33-47): element (i, j) = m[0][j]*this[i][0] + m[1][j]*this[i][1] + m[2][j]*this[i][2]
    multiply(m) {
        const a = this.buffer, b = m.buffer, r = new Array(9);
        for (let i = 0; i < 3; i++)
            for (let j = 0; j < 3; j++) r[3 * i + j] = b[j] * a[3 * i] + b[3 + j] * a[3 * i + 1] + b[6 + j] * a[3 * i + 2];
        return new Matrix3(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]);
    }
    clone() { const e = this.buffer; return new Matrix3(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8]); }
    static Eye(v) { v = v === undefined ? 1 : v; return new Matrix3(v, 0, 0, 0, v, 0, 0, 0, v); }
    static Diagonal(v) { return new Matrix3(v.x, 0, 0, 0, v.y, 0, 0, 0, v.z); }
    // src/math/Matrix3.ts:67-80
    static RotationFromQuaternion(q) {
        const x = q.x, y = q.y, z = q.z, w = q.w;
        return new Matrix3(
            1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
            2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
            2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y);
    }
    // src/math/Matrix3.ts:82-103
    static RotationFromEuler(m) {
        const cx = Math.cos(m.x), sx = Math.sin(m.x);
        const cy = Math.cos(m.y), sy = Math.sin(m.y);
        const cz = Math.cos(m.z), sz = Math.sin(m.z);
        return new Matrix3(
            cy * cz + sy * sx * sz, -cy * sz + sy * sx * cz, sy * cx,
            cx * sz, cx * cz, -sx,
            -sy * cz + cy * sx * sz, sy * sz + cy * sx * cz, cy * cx);
    }
}
module.exports = { Matrix3 };
