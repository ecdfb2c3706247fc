#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest default rc=$?"; tail -2 gpurun_out/pytest_gpu.log
BENCH_ARGS="--frames-in-flight 1" ENVS="" bash scripts/gpu_env_exp.sh
BENCH_ARGS="--frames-in-flight 1 --config C4 --steps 40 --warmup 5" ENVS="GSR_SEG_TARGET=8000" bash scripts/gpu_env_exp.sh
