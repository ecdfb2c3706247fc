#!/bin/bash
# full-size parity tests (C3, C4) + C4 bench
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_full.log
timeout -k 10 300 python bench.py --config C4 --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench c4 rc=$?"
timeout -k 10 300 python bench.py --config C2 --steps 240 --warmup 20 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "bench c2 rc=$?"
tail -2 gpurun_out/bench_c4.err
