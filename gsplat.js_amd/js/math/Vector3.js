"use strict";
// Immutable 3-vector, f64 components.  API of src/math/Vector3.ts.
class Vector3 {
    constructor(x, y, z) {
        this.x = x === undefined ? 0 : x;
        this.y = y === undefined ? 0 : y;
        this.z = z === undefined ? 0 : z;
    }
    equals(v) { return this.x === v.x && this.y === v.y && this.z === v.z; }
    _zip(v, f) {
        return typeof v === "number" ? new Vector3(f(this.x, v), f(this.y, v), f(this.z, v))
                                     : new Vector3(f(this.x, v.x), f(this.y, v.y), f(this.z, v.z));
    }
    add(v) { return this._zip(v, (a, b) => a + b); }
    subtract(v) { return this._zip(v, (a, b) => a - b); }
    multiply(v) { return this._zip(v, (a, b) => a * b); }
    lerp(v, t) { return new Vector3(this.x + (v.x - this.x) * t, this.y + (v.y - this.y) * t, this.z + (v.z - this.z) * t); }
    length() { return Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z); }
    distanceTo(v) { return Math.sqrt(Math.pow(this.x - v.x, 2) + Math.pow(this.y - v.y, 2) + Math.pow(this.z - v.z, 2)); }
    normalize() { const l = this.length(); return new Vector3(this.x / l, this.y / l, this.z / l); }
    flat() { return [this.x, this.y, this.z]; }
    clone() { return new Vector3(this.x, this.y, this.z); }
}
module.exports = { Vector3 };
