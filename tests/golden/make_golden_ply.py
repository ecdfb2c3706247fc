#!/usr/bin/env python3
"""tests/golden/ply_golden.json: two synthetic files (an INRIA .ply and a codebook-quantized one, oracle/ply_oracle.py's seeded
writers) and what the reference's OWN parsers return for them -- tests/golden/make_golden_ply.js cuts the three static parser
methods out of PLYLoader.ts at generation time, erases their type annotations and runs them under Node.
Run here:  python tests/golden/make_golden_ply.py"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ply_oracle as P  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    inria, quant = P.synth_ply(96, 31), P.synth_qply([40, 24, 16, 20], 37)
    a, b = os.path.join(d, "a.ply"), os.path.join(d, "q.ply")
    open(a, "wb").write(inria)
    open(b, "wb").write(quant)
    r = subprocess.run(["node", os.path.join(HERE, "make_golden_ply.js"), a, b], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    out = json.loads(r.stdout)
out["inria_ply"], out["quantized_ply"], out["quantized_counts"] = inria.hex(), quant.hex(), [40, 24, 16, 20]
json.dump(out, open(os.path.join(HERE, "ply_golden.json"), "w"))
print("wrote ply_golden.json:", len(inria), "+", len(quant), "input bytes; rows", len(out["plain"]) // 64, "/", len(out["q_rows"]) // 64, "bands", out["q_bands"])
