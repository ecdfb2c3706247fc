"use strict";
// src/renderers/webgl/passes/FadeInPass.ts: ramps u_depthFade from 0 to 1 in steps of speed*0.01 per frame and
// then switches the effect off.  The compositor has no depth-fade uniform yet (steady state = scalingFactor 1,
// vertex.glsl.ts:214-223; SURVEY.md 8(f) rank 3), so this keeps the hook and the counter only.
class FadeInPass {
    constructor(speed) {
        const step = (speed === undefined ? 1.0 : speed) * 0.01;
        this.value = 0.0;
        this.active = false;
        this.init = () => { this.value = 0; this.active = true; };
        this.render = () => {
            if (!this.active) return;
            this.value = Math.min(this.value + step, 1.0);
            if (this.value >= 1.0) this.active = false;
        };
    }
}
module.exports = { FadeInPass };
