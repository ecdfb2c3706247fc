"use strict";
// Headless OrbitControls: the pose mathematics of src/controls/OrbitControls.ts:264-283 without the DOM input
// handlers (there is no canvas to listen on).  alpha/beta/radius/target are set directly or through setCameraTarget;
// update() applies the same damped interpolation and writes camera.position / camera.rotation.
const { Vector3 } = require("../math/Vector3");
const { Quaternion } = require("../math/Quaternion");

class OrbitControls {
    constructor(camera, _domElement, alpha, beta, radius, enableKeyboardControls, inputTarget) {  // eslint-disable-line no-unused-vars
        this.minAngle = -90; this.maxAngle = 90; this.minZoom = 0.1; this.maxZoom = 30;
        this.orbitSpeed = 1; this.panSpeed = 1; this.zoomSpeed = 1; this.dampening = 0.12;
        let a = alpha === undefined ? 0.5 : alpha, b = beta === undefined ? 0.5 : beta, r = radius === undefined ? 5 : radius;
        let t = inputTarget || new Vector3();
        this.desiredAlpha = a; this.desiredBeta = b; this.desiredRadius = r; this.desiredTarget = t;
        const lerp = (x, y, k) => (1 - k) * x + k * y;
        this.setCameraTarget = (newTarget) => {
            const dx = newTarget.x - camera.position.x, dy = newTarget.y - camera.position.y, dz = newTarget.z - camera.position.z;
            this.desiredRadius = Math.sqrt(dx * dx + dy * dy + dz * dz);
            this.desiredBeta = Math.atan2(dy, Math.sqrt(dx * dx + dz * dz));
            this.desiredAlpha = -Math.atan2(dx, dz);
            this.desiredTarget = new Vector3(newTarget.x, newTarget.y, newTarget.z);
        };
        this.snap = () => { a = this.desiredAlpha; b = this.desiredBeta; r = this.desiredRadius; t = this.desiredTarget; };
        this.update = () => {
            a = lerp(a, this.desiredAlpha, this.dampening);
            b = lerp(b, this.desiredBeta, this.dampening);
            r = lerp(r, this.desiredRadius, this.dampening);
            t = t.lerp(this.desiredTarget, this.dampening);
            OrbitControls.applyPose(camera, a, b, r, t);
        };
        this.dispose = () => {};
        this.update();   // (OrbitControls.ts:351: the constructor leaves the camera on the orbit)
    }
    // OrbitControls.ts:275-283
    static applyPose(camera, alpha, beta, radius, target) {
        const x = target.x + radius * Math.sin(alpha) * Math.cos(beta);
        const y = target.y - radius * Math.sin(beta);
        const z = target.z - radius * Math.cos(alpha) * Math.cos(beta);
        camera.position = new Vector3(x, y, z);
        const d = target.subtract(camera.position).normalize();
        camera.rotation = Quaternion.FromEuler(new Vector3(Math.asin(-d.y), Math.atan2(d.x, d.z), 0));
    }
}
module.exports = { OrbitControls };
