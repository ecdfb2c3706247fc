#!/bin/bash
# frames/s through the Node host on the synthetic C3 scene: writes the scene with the Python generator, then tools/bench_node.js
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
CFG=${CFG:-C3}
python - <<PY || exit 1
import sys
sys.path[:0] = ["$GRAFT_REPO_ROOT", "$GRAFT_REPO_ROOT/gsplat.js_amd/py"]
import numpy as np, gsplat_hip as gh
rows = gh.synth.config_rows("$CFG")
np.ascontiguousarray(rows).tofile("/tmp/scene_$CFG.splat")
c = gh.synth.CONFIGS["$CFG"]
open("/tmp/scene_$CFG.args", "w").write("%d %d %s" % (c["width"], c["height"], c["fx"]))
PY
timeout -k 10 300 node tools/bench_node.js /tmp/scene_$CFG.splat $(cat /tmp/scene_$CFG.args) ${FRAMES:-480} 30 2> gpurun_out/bench_node_$CFG.err | tee gpurun_out/bench_node_$CFG.json
