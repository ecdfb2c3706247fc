"use strict";
// src/renderers/webgl/passes/FadeInPass.ts: init() switches u_useDepthFade on with u_depthFade = 0; every render()
// adds speed*0.01 to it until it reaches 1, then switches the effect off.  The "uniforms" live in the HIP renderer
// (gsr_set_depth_fade); the projection kernel applies vertex.glsl.ts:214-229.
class FadeInPass {
    constructor(speed) {
        const step = (speed === undefined ? 1.0 : speed) * 0.01;
        let value = 0.0, active = false, renderer = null;
        this.init = (r) => {
            value = 0;
            active = true;
            renderer = r;
            renderer.setDepthFade(true, value);
        };
        this.render = () => {
            if (!active) return;
            value = Math.min(value + step, 1.0);
            if (value >= 1.0) active = false;
            renderer.setDepthFade(active, value);
        };
        Object.defineProperty(this, "value", { get: () => value });
        Object.defineProperty(this, "active", { get: () => active });
    }
}
module.exports = { FadeInPass };
