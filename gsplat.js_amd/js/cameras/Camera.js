"use strict";
// Pinhole camera: API of src/cameras/Camera.ts.  update(w, h) builds the three matrices the hot path consumes
// (Camera.ts:81-92) in f64, column-major buffers; the renderer hands them over as Float32Array(buffer), which is
// where the reference rounds to f32 too (Worker.ts:37, WebGLRenderer.ts:147,159,275).
const { Object3D } = require("../core/Object3D");
const { Quaternion } = require("../math/Quaternion");
const { Matrix3 } = require("../math/Matrix3");
const { Matrix4 } = require("../math/Matrix4");
const { Vector3 } = require("../math/Vector3");

function vec3From(v) { return v.x !== undefined ? new Vector3(v.x, v.y, v.z) : new Vector3(v[0], v[1], v[2]); }
function quatFrom(r) {
    if (r.x !== undefined) return new Quaternion(r.x, r.y, r.z, r.w);
    const flat = Array.prototype.concat.apply([], Array.from(r));
    return Quaternion.FromMatrix3(new Matrix3(flat[0], flat[1], flat[2], flat[3], flat[4], flat[5], flat[6], flat[7], flat[8]));
}

class Camera extends Object3D {
    constructor(position, rotation, fx, fy, near, far) {
        super();
        this.position = position === undefined ? new Vector3(0, 0, 0) : position;
        this.rotation = rotation === undefined ? new Quaternion() : rotation;
        this.fx = fx === undefined ? 1132 : fx;
        this.fy = fy === undefined ? 1132 : fy;
        this.near = near === undefined ? 0.01 : near;
        this.far = far === undefined ? 1000 : far;
        this.projectionMatrix = new Matrix4();
        this.viewMatrix = new Matrix4();
        this.viewProj = new Matrix4();
        this.viewToWorld = new Matrix4();
    }

    // Camera.ts:81-92 (+ getViewMatrix :32-56)
    update(width, height) {
        const f = this.far, n = this.near;
        this.projectionMatrix = new Matrix4(
            2 * this.fx / width, 0, 0, 0,
            0, -2 * this.fy / height, 0, 0,
            0, 0, f / (f - n), 1,
            0, 0, -(f * n) / (f - n), 0);
        const R = Matrix3.RotationFromQuaternion(this.rotation).buffer;
        const t = this.position.flat();
        this.viewToWorld = new Matrix4(R[0], R[3], R[6], t[0], R[1], R[4], R[7], t[1], R[2], R[5], R[8], t[2], 0, 0, 0, 1);
        this.viewMatrix = new Matrix4(
            R[0], R[1], R[2], 0,
            R[3], R[4], R[5], 0,
            R[6], R[7], R[8], 0,
            -t[0] * R[0] - t[1] * R[3] - t[2] * R[6],
            -t[0] * R[1] - t[1] * R[4] - t[2] * R[7],
            -t[0] * R[2] - t[1] * R[5] - t[2] * R[8],
            1);
        this.viewProj = this.projectionMatrix.multiply(this.viewMatrix);
    }

    // camera JSON import/export (Camera.ts:95-181); no console chatter
    setFromData(data) {
        this.rotation = quatFrom(data.rotation);
        this.position = vec3From(data.position);
        this.fx = data.fx;
        this.fy = data.fy;
        this.update(data.width, data.height);
    }
    static fromData(data) {
        const cam = new Camera(vec3From(data.position), quatFrom(data.rotation), data.fx, data.fy);
        cam.update(data.width, data.height);
        return cam;
    }
    dumpSettings(width, height) {
        return { id: 0, img_name: "NONE", width: width, height: height, position: this.position, rotation: this.rotation,
                 fy: this.fy, fx: this.fx };
    }
}
module.exports = { Camera };
