#!/bin/bash
# quick: parity tests + exact/early-out bench + kernel trace (no sweeps)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
rm -f gpurun_out/bench1_seg*.json
timeout -k 10 300 python bench.py --steps 240 --warmup 20 $BENCH_ARGS > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 240 --warmup 20 --early-out-eps 1e-4 --no-cpu-baseline > gpurun_out/bench1_eo.json 2>> gpurun_out/bench1.err; echo "bench eo rc=$?"
rm -rf gpurun_out/prof1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --timed-only > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1; echo "prof rc=$?"
