#!/bin/bash
# round 2: new tests (RCCL self test, JS group), gloo self-launch rehearsal
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rccl or allgather or overflow or sort_host or limit_box or ideal or closed_form or bands or js" > gpurun_out/r2_pytest_new.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -15 gpurun_out/r2_pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r2_bench_gloo2.json 2> gpurun_out/r2_bench_gloo2.err; echo "bench gloo2 rc=$?"
tail -3 gpurun_out/r2_bench_gloo2.err; head -c 600 gpurun_out/r2_bench_gloo2.json
