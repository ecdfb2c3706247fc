#!/bin/bash
# parity tests against an experiment library, then A/B bench: LIB=fused bash scripts/gpu_ab.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
GSPLAT_HIP_LIB=$GRAFT_REPO_ROOT/gsplat.js_amd/lib_exp/$LIB/libgsplat_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_$LIB.log 2>&1; echo "pytest $LIB rc=$?"; tail -3 gpurun_out/pytest_$LIB.log
BENCH_ARGS="--frames-in-flight 1" bash scripts/gpu_exp.sh
BENCH_ARGS="--steps 480" bash scripts/gpu_exp.sh
BENCH_ARGS="--frames-in-flight 1 --config C4 --steps 40 --warmup 5" bash scripts/gpu_exp.sh
