"use strict";
// Public API: the export list of src/index.ts:1-12.  WebGLRenderer is the HIP renderer under the name callers of
// the reference already use.
const { HIPRenderer, sortHost } = require("./renderers/HIPRenderer");
module.exports = {
    Camera: require("./cameras/Camera").Camera,
    Scene: require("./core/Scene").Scene,
    Loader: require("./loaders/Loader").Loader,
    PLYLoader: require("./loaders/PLYLoader").PLYLoader,
    WebGLRenderer: HIPRenderer,
    HIPRenderer: HIPRenderer,
    OrbitControls: require("./controls/OrbitControls").OrbitControls,
    Quaternion: require("./math/Quaternion").Quaternion,
    Vector3: require("./math/Vector3").Vector3,
    Matrix4: require("./math/Matrix4").Matrix4,
    Matrix3: require("./math/Matrix3").Matrix3,
    ShaderPass: require("./renderers/ShaderPass").ShaderPass,
    FadeInPass: require("./renderers/FadeInPass").FadeInPass,
    sortHost: sortHost,
};
