"use strict";
// Plug-in hook of src/renderers/webgl/passes/ShaderPass.ts: init(renderer, program) once per (re)initialisation,
// render() once per frame.  There is no GL program here; `program` is null.
class ShaderPass {
    init(renderer, program) {}  // eslint-disable-line no-unused-vars
    render() {}
}
module.exports = { ShaderPass };
