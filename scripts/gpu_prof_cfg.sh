#!/bin/bash
# kernel trace of one bench configuration: ARGS="--config C4 --steps 30 --warmup 5 --frames-in-flight 1" bash scripts/gpu_prof_cfg.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -rf gpurun_out/profc
export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $ARGS > $GRAFT_REPO_ROOT/gpurun_out/profc.log 2>&1; echo "prof rc=$?"
tail -1 $GRAFT_REPO_ROOT/gpurun_out/profc.log | cut -c1-300
