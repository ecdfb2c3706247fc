#!/bin/bash
# round 2 evidence for profiles/: kernel traces (one frame in flight / default), PMC passes, bench lines C3 C2 C4
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02 && rm -rf gpurun_out/r02/* gpurun_out/pmc
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'gsplat.js_amd/py'); import gsplat_hip as g; print(g.build_id())" > gpurun_out/r02/build_id.txt 2>/dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/trace_1inflight -- python3 $GRAFT_REPO_ROOT/bench.py --timed-only --frames-in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r02/trace_1inflight.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/trace_1inflight.err; echo "trace1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/trace_3inflight -- python3 $GRAFT_REPO_ROOT/bench.py --timed-only > $GRAFT_REPO_ROOT/gpurun_out/r02/trace_3inflight.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/trace_3inflight.err; echo "trace3 rc=$?"
cd $GRAFT_REPO_ROOT
bash scripts/gpu_pmc.sh
cp gpurun_out/r02/build_id.txt gpurun_out/pmc/build_id.txt
python scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/r02/pmc_summary.txt 2>&1
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err; echo "bench c3 rc=$?"
timeout -k 10 300 python bench.py --early-out-eps 1e-4 --no-cpu-baseline > gpurun_out/r02/bench_c3_earlyout.json 2>/dev/null; echo "bench eo rc=$?"
timeout -k 10 300 python bench.py --config C2 --no-cpu-baseline > gpurun_out/r02/bench_c2.json 2>/dev/null; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --config C4 --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/r02/bench_c4.json 2>/dev/null; echo "bench c4 rc=$?"
ls gpurun_out/r02
