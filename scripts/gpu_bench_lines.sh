#!/bin/bash
# the bench lines alone (after profiles/blend_traffic.json has been regenerated for this build)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err; echo "bench c3 rc=$?"
timeout -k 10 300 python bench.py --early-out-eps 1e-4 --no-cpu-baseline > gpurun_out/r02/bench_c3_earlyout.json 2>/dev/null; echo "bench eo rc=$?"
timeout -k 10 300 python bench.py --config C2 --no-cpu-baseline > gpurun_out/r02/bench_c2.json 2>/dev/null; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --config C4 --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/r02/bench_c4.json 2>/dev/null; echo "bench c4 rc=$?"
