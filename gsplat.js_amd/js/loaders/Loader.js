"use strict";
// .splat reader for Node (src/loaders/Loader.ts fetches over HTTP in a browser; here the bytes come from fs).
const fs = require("fs");
const { Scene } = require("../core/Scene");

class Loader {
    // same name/arity as Loader.LoadAsync(url, scene, onProgress): `url` is a file path here
    static async LoadAsync(file, scene, onProgress) {
        const buf = await fs.promises.readFile(file);
        if (onProgress) onProgress(1);
        return Loader._apply(buf, scene);
    }
    static LoadSync(file, scene) { return Loader._apply(fs.readFileSync(file), scene); }
    static async LoadFromFileAsync(file, scene, onProgress) { return Loader.LoadAsync(file, scene, onProgress); }
    static _apply(buf, scene) {
        if (buf.length % Scene.RowLength) throw new Error("not a .splat file: length is not a multiple of " + Scene.RowLength);
        const bytes = new Uint8Array(buf.length);   // aligned private copy (Buffers are pooled at odd offsets)
        bytes.set(buf);
        scene.setData(bytes);
        return scene;
    }
}
module.exports = { Loader };
