#!/bin/bash
# round 2 first GPU pass: full gpu test suite, default bench
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -15 gpurun_out/r2_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err; echo "bench rc=$?"
tail -3 gpurun_out/r2_bench_a.err
