#!/usr/bin/env python3
"""Generate tests/golden/sort_*.json|npz from the REFERENCE's own code.

Run in the build container (needs /root/reference): `python tests/golden/make_golden.py`.
The reference's wasm/wasm.cpp is compiled from source where it lies (oracle/Makefile ->
oracle/_ref/libref_sort.so, never copied into this repo) and driven through
oracle.ref_sort(), which presets starts[65536] as SURVEY.md 8(c) defines.  What is
committed is data only: scene seeds, camera matrices, and the resulting depthIndex
arrays (in full for small N, sha256 + head/tail for large N).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))

from gsplat_hip import synth, camera  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [  # (name, n, seed, poses)  G1 of SURVEY 8(c); C1 doubles as G3 (N < 65536)
    ("g1_4096", 4096, 1, (0, 17, 63)),
    ("g3_c1_10000", 10000, 1, (0, 40, 95)),
    ("g1_70000", 70000, 2, (0, 17, 63)),
    ("g1_1000000", 1000000, 3, (0, 17, 63)),
]


def tie_case():
    """G2: many exact ties and >= 3 splats tied at maxDepth (key 65536)."""
    rng = np.random.default_rng(7)
    n = 5000
    pos = rng.integers(-8, 9, size=(n, 3)).astype(np.float32) * 0.25
    pos[[10, 200, 4000]] = [0.0, 0.0, 50.0]
    vp = np.zeros(16, dtype=np.float32)
    vp[10] = 1.0
    return pos, vp


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert O.ref_available(), "oracle/_ref is not built (needs /root/reference)"
    meta = {"generator": "tests/golden/make_golden.py", "source": "wasm/wasm.cpp compiled with g++ -O2 -ffp-contract=off", "cases": []}
    full = {}
    for name, n, seed, poses in CASES:
        rows = synth.synth_rows(n, seed)
        _, pos = O.scene_pack(rows)
        for k in poses:
            vp = camera.orbit_camera(k).f32()[2]
            di, keys = O.ref_sort(vp, pos, calls=3)  # 3 back-to-back calls: state carried in starts[] is handled
            assert np.array_equal(np.sort(di), np.arange(n, dtype=np.uint32)), "reference output is not a permutation"
            entry = {"name": name, "n": n, "seed": seed, "pose": k, "viewProj": [float(x) for x in vp],
                     "rows_sha256": sha(rows), "depthIndex_sha256": sha(di), "keys_sha256": sha(keys),
                     "n_max_bucket": int((keys == 65536).sum()),
                     "head": di[:64].tolist(), "tail": di[-64:].tolist()}
            if n <= 10000:
                full["%s_p%d" % (name, k)] = di
            meta["cases"].append(entry)
    pos, vp = tie_case()
    di, keys = O.ref_sort(vp, pos, calls=2)
    meta["tie_case"] = {"depthIndex_sha256": sha(di), "keys_sha256": sha(keys), "n_max_bucket": int((keys == 65536).sum())}
    full["g2_ties"] = di
    json.dump(meta, open(os.path.join(HERE, "sort_golden.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "sort_golden_full.npz"), **full)
    print("wrote %d cases" % len(meta["cases"]))


if __name__ == "__main__":
    main()
