// TypeScript surface of the MI355X drop-in.  Mirrors the declarations a user of Lanv1/gsplat.js imports
// (src/index.ts:1-12); WebGLRenderer is the HIP renderer.
export class Vector3 {
    readonly x: number; readonly y: number; readonly z: number;
    constructor(x?: number, y?: number, z?: number);
    equals(v: Vector3): boolean;
    add(v: Vector3 | number): Vector3;
    subtract(v: Vector3 | number): Vector3;
    multiply(v: Vector3 | number): Vector3;
    lerp(v: Vector3, t: number): Vector3;
    length(): number;
    distanceTo(v: Vector3): number;
    normalize(): Vector3;
    flat(): number[];
    clone(): Vector3;
}
export class Quaternion {
    readonly x: number; readonly y: number; readonly z: number; readonly w: number;
    constructor(x?: number, y?: number, z?: number, w?: number);
    equals(q: Quaternion): boolean;
    normalize(): Quaternion;
    multiply(q: Quaternion): Quaternion;
    flat(): number[];
    clone(): Quaternion;
    toEuler(): Vector3;
    static FromEuler(e: Vector3): Quaternion;
    static FromMatrix3(m: Matrix3): Quaternion;
}
export class Matrix3 {
    readonly buffer: number[];
    constructor(n11?: number, n12?: number, n13?: number, n21?: number, n22?: number, n23?: number, n31?: number, n32?: number, n33?: number);
    equals(m: Matrix3): boolean;
    multiply(m: Matrix3): Matrix3;
    clone(): Matrix3;
    static Eye(v?: number): Matrix3;
    static Diagonal(v: Vector3): Matrix3;
    static RotationFromQuaternion(q: Quaternion): Matrix3;
    static RotationFromEuler(m: Vector3): Matrix3;
}
export class Matrix4 {
    readonly buffer: number[];
    constructor(...n: number[]);
    equals(m: Matrix4): boolean;
    multiply(m: Matrix4): Matrix4;
    clone(): Matrix4;
}
export interface SceneEvent { type: string }
export class Scene {
    static RowLength: number;
    constructor();
    addEventListener(type: string, listener: (e: SceneEvent) => void): void;
    removeEventListener(type: string, listener: (e: SceneEvent) => void): void;
    hasEventListener(type: string, listener: (e: SceneEvent) => void): boolean;
    dispatchEvent(e: SceneEvent): void;
    setData(data: Uint8Array, shs?: Float32Array): void;
    translate(translation: Vector3): void;
    rotate(rotation: Quaternion): void;
    scale(scale: Vector3): void;
    limitBox(xMin: number, xMax: number, yMin: number, yMax: number, zMin: number, zMax: number): void;
    saveToFile(name: string): void;
    toSplatBytes(): Uint8Array;
    data: Uint32Array; vertexCount: number; width: number; height: number;
    positions: Float32Array; rotations: Float32Array; scales: Float32Array;
    shs: Uint32Array; shs_rgb: [Uint32Array, Uint32Array, Uint32Array]; shHeight: number;
    g0bands: number; bandsIndices: Int32Array;
}
export class Camera {
    position: Vector3; rotation: Quaternion;
    fx: number; fy: number; near: number; far: number;
    projectionMatrix: Matrix4; viewMatrix: Matrix4; viewProj: Matrix4; viewToWorld: Matrix4;
    constructor(position?: Vector3, rotation?: Quaternion, fx?: number, fy?: number, near?: number, far?: number);
    update(width: number, height: number): void;
    setFromData(data: any): void;
    static fromData(data: any): Camera;
    dumpSettings(width: number, height: number): object;
    addEventListener(type: string, listener: (e: SceneEvent) => void): void;
    removeEventListener(type: string, listener: (e: SceneEvent) => void): void;
}
export class ShaderPass { init(renderer: HIPRenderer, program: null): void; render(): void; }
export class FadeInPass implements ShaderPass {
    constructor(speed?: number);
    init(renderer: HIPRenderer, program: null): void;
    render(): void;
}
export interface HIPRendererOptions {
    width?: number; height?: number; device?: number;
    /** 0 (default): composite every splat like the reference; >0: a tile stops once every pixel's 1-alpha is below this */
    earlyOutEps?: number;
    /** multi-GPU: composite only pixel columns [x0, x1) */
    band?: [number, number];
    timing?: boolean;
    /** several renderers keep frames in flight on one device (GSR_FLAG_THROUGHPUT): longer compositor work items */
    throughput?: boolean;
}
export interface FrameStats {
    msProjectKey: number; msSort: number; msBin: number; msBlend: number; msCombine: number; msTotal: number;
    visible: number; binEntries: number; tileEntries: number; n: number; frames: number;
}
export class HIPRenderer {
    width: number; height: number;
    constructor(targetOrOptions?: HIPRendererOptions | { width: number; height: number } | null, shaderPasses?: ShaderPass[] | null);
    render(scene: Scene, camera: Camera): void;
    /** enqueue the frame and return; pair with sync() (several `throughput` renderers used round-robin keep the GPU full) */
    renderAsync(scene: Scene, camera: Camera): void;
    /** waits for the enqueued frames; throws once if asynchronous frames were lost to a list overflow (the renderer stays usable) */
    sync(): void;
    overflowPending(): boolean;
    setListCapacity(entries: number): void;
    /** multi-GPU, one process per GPU. Collective: same id (createGroupId() on rank 0), world and edges on every rank.
     *  Afterwards render() draws this rank's tile-column band and all-gathers the RGBA8 frame over xGMI inside the
     *  library (RCCL); readPixels() returns the whole frame on every rank. */
    joinGroup(group: { id: Uint8Array; rank: number; world: number; edges: Array<[number, number]> }): void;
    /** Another renderer of the same rank (frames in flight) joins the group `leader` has joined: one communicator and one
     *  exchange stream per rank (gsr_comm_share).  Leave (or dispose) it before the leader. */
    shareGroup(leader: HIPRenderer): void;
    leaveGroup(): void;
    group(): { rank: number; world: number } | null;
    static createGroupId(): Uint8Array;
    static bandEdges(width: number, world: number): Array<[number, number]>;
    sort(camera?: Camera): void;
    setSize(width: number, height: number): void;
    resize(): void;
    setBand(x0: number, x1: number): void;
    setCameraBuffers(): void;
    setShTextures(): void;
    setDepthFade(useDepthFade: boolean, depthFade: number): void;
    /** device-side scene: .splat rows in; Scene.setData and the transforms run as kernels, bit-identical to Scene */
    setSceneRows(rows: Uint8Array): void;
    sceneTranslate(t: Vector3): void;
    sceneRotate(q: Quaternion): void;
    sceneScale(s: Vector3): void;
    sceneLimitBox(xMin: number, xMax: number, yMin: number, yMax: number, zMin: number, zMax: number): void;
    readSceneData(): { data: Uint32Array; positions: Float32Array; vertexCount: number };
    renderDeviceScene(camera: Camera): void;
    dispose(): void;
    /** RGBA8, row 0 = top, round(clamp(x,0,1)*255), premultiplied alpha */
    /** RGBA8, row 0 = top; pass an array of width*height*4 elements to have it filled and returned (no allocation per frame). */
    readPixels(out?: Uint8Array): Uint8Array;
    readPixelsFloat(out?: Float32Array): Float32Array;
    lastDepthIndex(): Uint32Array;
    stats(): FrameStats;
    deviceInfo(): { name: string; computeUnits: number; clockKhz: number };
}
export { HIPRenderer as WebGLRenderer };
export class Loader {
    static LoadAsync(file: string, scene: Scene, onProgress?: (p: number) => void): Promise<Scene>;
    static LoadFromFileAsync(file: string, scene: Scene, onProgress?: (p: number) => void): Promise<Scene>;
    static LoadSync(file: string, scene: Scene): Scene;
}
export class PLYLoader {
    /** format: "" | "polycam"; useShs: also read the 45 f_rest_* floats; quantized (with useShs): the codebook variant */
    static LoadAsync(file: string, scene: Scene, onProgress?: (p: number, done?: boolean) => void, format?: string,
                     useShs?: boolean, quantized?: boolean): Promise<Scene>;
    static LoadFromFileAsync(file: string, scene: Scene, onProgress?: (p: number, done?: boolean) => void, format?: string,
                             useShs?: boolean, quantized?: boolean): Promise<Scene>;
    static LoadFromBytes(bytes: Uint8Array, scene: Scene, format?: string, useShs?: boolean, quantized?: boolean): Scene;
}
export class OrbitControls {
    minAngle: number; maxAngle: number; minZoom: number; maxZoom: number;
    orbitSpeed: number; panSpeed: number; zoomSpeed: number; dampening: number;
    desiredAlpha: number; desiredBeta: number; desiredRadius: number; desiredTarget: Vector3;
    constructor(camera: Camera, domElement?: unknown, alpha?: number, beta?: number, radius?: number,
                enableKeyboardControls?: boolean, inputTarget?: Vector3);
    setCameraTarget(newTarget: Vector3): void;
    snap(): void;
    update(): void;
    dispose(): void;
    static applyPose(camera: Camera, alpha: number, beta: number, radius: number, target: Vector3): void;
}
/** Drop-in for the wasm export `sort` (wasm/wasm.cpp:8-13) on host typed arrays. */
export function sortHost(viewProj: Float32Array, vertexCount: number, fBuffer: Float32Array,
                         depthBuffer: Uint32Array | null, depthIndex: Uint32Array): void;
