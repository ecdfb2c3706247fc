#!/bin/bash
# kernel trace of an arbitrary python script: SCRIPT="scripts/big_scene_check.py 20000000" bash scripts/gpu_prof_py.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -rf gpurun_out/profp
export TMPDIR=/tmp
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profp -- python3 $GRAFT_REPO_ROOT/$SCRIPT > $GRAFT_REPO_ROOT/gpurun_out/profp.log 2>&1; echo "prof rc=$?"
tail -2 $GRAFT_REPO_ROOT/gpurun_out/profp.log | cut -c1-300
