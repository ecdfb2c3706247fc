"use strict";
// position/rotation holder that fires "change" (src/core/Object3D.ts).
const { EventDispatcher } = require("./EventDispatcher");
const { Vector3 } = require("../math/Vector3");
const { Quaternion } = require("../math/Quaternion");

class Object3D extends EventDispatcher {
    constructor() {
        super();
        this._position = new Vector3();
        this._rotation = new Quaternion();
    }
    get position() { return this._position; }
    set position(p) { if (!this._position.equals(p)) { this._position = p; this.dispatchEvent({ type: "change" }); } }
    get rotation() { return this._rotation; }
    set rotation(r) { if (!this._rotation.equals(r)) { this._rotation = r; this.dispatchEvent({ type: "change" }); } }
}
module.exports = { Object3D };
