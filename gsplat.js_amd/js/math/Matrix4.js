"use strict";
// 4x4, f64.  `buffer` is what the reference uploads with transpose=false, i.e. buffer[c*4+r] = element
// (row r, column c).  API of src/math/Matrix4.ts.
class Matrix4 {
    constructor() {
        const ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1];
        this.buffer = ident.map((v, i) => (arguments[i] === undefined ? v : arguments[i]));
    }
    equals(m) {
        if (this.buffer === m.buffer) return true;
        if (this.buffer.length !== m.buffer.length) return false;
        return this.buffer.every((v, i) => v === m.buffer[i]);
    }
    // src/math/Matrix4.ts:32-53: out[4i+j] = m[4i]*this[j] + m[4i+1]*this[4+j] + m[4i+2]*this[8+j] + m[4i+3]*this[12+j]
    multiply(m) {
        const a = this.buffer, b = m.buffer, r = new Array(16);
        for (let i = 0; i < 4; i++)
            for (let j = 0; j < 4; j++)
                r[4 * i + j] = b[4 * i] * a[j] + b[4 * i + 1] * a[4 + j] + b[4 * i + 2] * a[8 + j] + b[4 * i + 3] * a[12 + j];
        return Matrix4.fromArray(r);
    }
    clone() { return Matrix4.fromArray(this.buffer); }
    static fromArray(a) { const m = new Matrix4(); for (let i = 0; i < 16; i++) m.buffer[i] = a[i]; return m; }
}
module.exports = { Matrix4 };
