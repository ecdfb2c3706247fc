"use strict";
// HIPRenderer: drop-in for the render path of src/renderers/WebGLRenderer.ts on an AMD MI355X.
//   renderer.render(scene, camera) = camera.update + depth sort + projection + front-to-back composite,
// all on one HIP stream through libgsplat_hip.so (see include/gsplat_hip.h).  Differences from the WebGL
// renderer, all forced by the missing browser: no `domElement`/`gl`; the image is read back with readPixels() /
// readPixelsFloat(); the sort is synchronous with the frame (the reference sorts in a worker and draws with a
// stale order until it finishes, WebGLRenderer.ts:105-110,223-229), so every frame is deterministic.
// There is no CPU fallback: the constructor throws if the addon or a GPU is missing.
const path = require("path");
const { FadeInPass } = require("./FadeInPass");

let native = null;
function loadNative() {
    if (!native) {
        try {
            native = require(path.join(__dirname, "..", "native", "gsplat_hip.node"));
        } catch (e) {
            throw new Error("gsplat_hip.node is not built or libgsplat_hip.so cannot be loaded (" + e.message +
                            "); run `python -c \"import __graft_entry__ as g; g.build()\"`");
        }
    }
    return native;
}

class HIPRenderer {
    // new HIPRenderer(canvasLike | options | null, shaderPasses | null)
    //   canvasLike: anything with numeric width/height (stands in for the HTMLCanvasElement of WebGLRenderer.ts:23)
    //   options: { width, height, device, earlyOutEps, band: [x0, x1], timing, throughput }
    //   (throughput: several renderers keep frames in flight on one device; see GSR_FLAG_THROUGHPUT)
    constructor(target, optionalShaderPasses) {
        const o = target || {};
        this._n = loadNative();
        this.width = o.width || 1920;
        this.height = o.height || 1080;
        const band = o.band || [0, 0];
        this._h = this._n.create({ device: o.device || 0, width: this.width, height: this.height,
                                   earlyOutEps: o.earlyOutEps || 0, bandX0: band[0], bandX1: band[1], timing: o.timing ? 1 : 0, throughput: o.throughput ? 1 : 0 });
        const passes = optionalShaderPasses || [];
        if (!optionalShaderPasses) passes.push(new FadeInPass());
        let activeScene = null, activeCamera = null, initialized = false, vertexCount = 0;
        const f32 = { view: new Float32Array(16), proj: new Float32Array(16), vp: new Float32Array(16) };

        const upload = () => {   // initWebGL's scene part: worker init + texImage2D (WebGLRenderer.ts:105-110,185-195)
            vertexCount = activeScene.vertexCount;
            this._n.setScene(this._h, activeScene.data, activeScene.positions, vertexCount);
            this.setShTextures();
            for (const p of passes) p.init(this, null);
            initialized = true;
        };
        const onSceneChange = () => upload();   // WebGLRenderer.ts:234-239

        this.setSize = (width, height) => {     // WebGLRenderer.ts:85-102
            this.width = width;
            this.height = height;
            this._n.resize(this._h, width, height);
        };
        this.resize = () => {};                  // no DOM: nothing to measure
        this.setBand = (x0, x1) => this._n.setBand(this._h, x0, x1);

        const pushCamera = () => {               // uniforms + postMessage({viewProj}) (WebGLRenderer.ts:144-159,268-269,275)
            activeCamera.update(this.width, this.height);
            f32.view.set(activeCamera.viewMatrix.buffer);         // f64 -> f32 exactly like new Float32Array(m.buffer)
            f32.proj.set(activeCamera.projectionMatrix.buffer);
            f32.vp.set(activeCamera.viewProj.buffer);
            this._n.setCamera(this._h, f32.view, f32.proj, f32.vp, activeCamera.fx, activeCamera.fy);
        };
        this.setCameraBuffers = () => pushCamera();
        // the two uniforms a FadeInPass drives (u_useDepthFade, u_depthFade)
        this.setDepthFade = (use, value) => this._n.setDepthFade(this._h, use ? 1 : 0, value);
        // SH textures + u_bandIndex, only for scenes that carry SH data (WebGLRenderer.ts:202-211,321-366)
        this.setShTextures = () => {
            if (!activeScene || !activeScene.shHeight) return;
            const band = activeScene.bandsIndices;
            const t = activeScene.shs_rgb;
            this._n.setSceneSh(this._h, t[0], t[1], t[2], activeScene.vertexCount - (band[0] + 1), band);
        };

        // ---- multi-GPU (one process per GPU): joinGroup() is collective -- every rank calls it with the same id
        // (HIPRenderer.createGroupId() on rank 0, passed on by the host: a file, a socket, an env var), its rank, the
        // world size and the same band edges [[x0, x1], ...] (contiguous runs of whole 32-px columns covering the
        // width; HIPRenderer.bandEdges(width, world) gives equal ones).  After that render(scene, camera) draws this
        // rank's band and all-gathers the RGBA8 slabs over xGMI (RCCL) inside the library, and readPixels() returns the
        // whole frame on every rank.
        let group = null;
        this.joinGroup = (g) => {
            const x0 = new Int32Array(g.world), x1 = new Int32Array(g.world);
            for (let q = 0; q < g.world; q++) { x0[q] = g.edges[q][0]; x1[q] = g.edges[q][1]; }
            this._n.commInit(this._h, g.id, g.rank, g.world, x0, x1);
            group = { rank: g.rank, world: g.world };
        };
        // A second renderer of the SAME rank (frames in flight with renderAsync) joins through the first one: it shares the
        // communicator and the exchange stream, so the rank's collectives go out in frame order (gsr_comm_share).
        this.shareGroup = (leader) => {
            this._n.commShare(this._h, leader._h);
            group = Object.assign({}, leader.group());
        };
        this.leaveGroup = () => { this._n.commDestroy(this._h); group = null; };
        this.group = () => group;

        // WebGLRenderer.ts:241-296
        this.render = (scene, camera) => {
            activeCamera = camera;
            if (scene !== activeScene) {
                if (activeScene) activeScene.removeEventListener("change", onSceneChange);
                activeScene = scene;
                activeScene.addEventListener("change", onSceneChange);
                upload();
            }
            pushCamera();
            for (const p of passes) p.render();
            if (group) {
                // band frame, then the framebuffer all-gather; readPixels() waits for the exchange.  The frame is rendered with
                // the blocking call, which repairs a bin-list overflow (regrow + render again) before the band is packed:
                // behind renderAsync the host would not know yet that the compositor drew nothing, and ship the old band.
                this._n.render(this._h);
                this._n.allgatherFrameAsync(this._h);
            } else {
                this._n.render(this._h);
            }
        };
        // Frames in flight (no counterpart in the reference, whose render() is one synchronous draw): renderAsync()
        // enqueues the frame and returns; sync() waits for it.  Several renderers created with { throughput: true }
        // and used round-robin keep the GPU full (bench.py --frames-in-flight).
        this.renderAsync = (scene, camera) => {
            activeCamera = camera;
            if (scene !== activeScene) {
                if (activeScene) activeScene.removeEventListener("change", onSceneChange);
                activeScene = scene;
                activeScene.addEventListener("change", onSceneChange);
                upload();
            }
            pushCamera();
            for (const p of passes) p.render();
            this._n.renderAsync(this._h);
            if (group) this._n.allgatherFrameAsync(this._h);
        };
        // sync() throws once (code GSPLAT_HIP, "... frame(s) were not composited") when asynchronous frames were lost
        // to a list overflow; the renderer stays usable and the last frame has been rendered again.
        this.sync = () => this._n.sync(this._h);
        this.overflowPending = () => this._n.overflowPending(this._h);
        this.setListCapacity = (entries) => this._n.setListCapacity(this._h, entries);
        this.sort = (camera) => {                // the worker's job alone (Worker.ts:36-43)
            if (camera) { activeCamera = camera; pushCamera(); }
            this._n.sort(this._h);
        };
        this.dispose = () => {                   // WebGLRenderer.ts:298-310
            if (activeScene) activeScene.removeEventListener("change", onSceneChange);
            activeScene = null;
            if (this._h) { this._n.destroy(this._h); this._h = null; }
            initialized = false;
        };

        // ---- optional device-side scene (SURVEY 8(f) rank 2): .splat rows in, Scene.setData and the transforms run as
        // kernels (bit-identical to Scene.js), no re-upload per change; render with renderDeviceScene(camera) ----
        this.setSceneRows = (rows) => {
            if (activeScene) activeScene.removeEventListener("change", onSceneChange);
            activeScene = null;
            this._n.setSceneRows(this._h, rows);
            vertexCount = rows.length / 32;
            for (const p of passes) p.init(this, null);
            initialized = true;
        };
        const xf = (kind, args) => { vertexCount = this._n.sceneTransform(this._h, kind, new Float64Array(args)); };
        this.sceneTranslate = (v) => xf(0, [v.x, v.y, v.z]);
        this.sceneRotate = (q) => xf(1, [q.x, q.y, q.z, q.w]);
        this.sceneScale = (v) => xf(2, [v.x, v.y, v.z]);
        this.sceneLimitBox = (xMin, xMax, yMin, yMax, zMin, zMax) => xf(3, [xMin, xMax, yMin, yMax, zMin, zMax]);
        this.readSceneData = () => {
            const data = new Uint32Array(vertexCount * 8), positions = new Float32Array(vertexCount * 3);
            this._n.readScene(this._h, data, positions);
            return { data: data, positions: positions, vertexCount: vertexCount };
        };
        this.renderDeviceScene = (camera) => {
            activeCamera = camera;
            pushCamera();
            for (const p of passes) p.render();
            this._n.render(this._h);
        };

        // ---- results ----
        this.lastDepthIndex = () => { const a = new Uint32Array(vertexCount); this._n.readDepthIndex(this._h, a); return a; };
        // readPixels(out?) / readPixelsFloat(out?): like gl.readPixels, a caller that reads every frame passes its own
        // array (width*height*4 elements) and gets it back filled; without one a fresh array is allocated per call, which
        // costs more than the copy itself (8 MB of zeroed pages at 1080p: 583 -> frames/s with a reused array in
        // tools/bench_node.js).
        this.readPixels = (out) => {
            const a = out || new Uint8Array(this.width * this.height * 4);
            if (group) this._n.readFrame(this._h, a, this.width, this.height);   // the gathered frame of all ranks
            else this._n.readPixels(this._h, a, this.width, this.height);
            return a;
        };
        this.readPixelsFloat = (out) => { const a = out || new Float32Array(this.width * this.height * 4); this._n.readPixels(this._h, a, this.width, this.height); return a; };
        this.stats = () => this._n.getTimings(this._h);
        this.deviceInfo = () => this._n.deviceInfo(this._h);
        this.isInitialized = () => initialized;
    }
}

HIPRenderer.createGroupId = () => loadNative().commUniqueId();
// equal bands of whole 32-px compositor columns, the tail clipped to the image (same rule as gsplat_hip.bands.band_edges)
HIPRenderer.bandEdges = (width, world) => {
    const nbx = Math.ceil(width / 32), per = Math.ceil(nbx / world), e = [];
    for (let q = 0; q < world; q++) e.push([Math.min(q * per * 32, width), Math.min((q + 1) * per * 32, width)]);
    return e;
};

// the wasm export's drop-in (wasm/wasm.cpp:8-13, call site Worker.ts:39)
function sortHost(viewProj, vertexCount, fBuffer, depthBuffer, depthIndex) {
    loadNative().sortHost(viewProj, vertexCount, fBuffer, depthBuffer || null, depthIndex);
}

module.exports = { HIPRenderer, sortHost };
