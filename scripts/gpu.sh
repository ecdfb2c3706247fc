#!/bin/bash
# The one GPU-box runner: `gpurun -- bash scripts/gpu.sh TASK [ARGS...] [-- TASK ...]`; output goes to gpurun_out/.
#   test [PYTEST_ARGS]        pytest -m gpu (one process) -> gpurun_out/pytest_gpu.log
#   bench NAME [BENCH_ARGS]   bench.py BENCH_ARGS -> gpurun_out/bench_NAME.json
#   ab ARGS...                scripts/ab_bench.py ARGS (variants "base@ENV=v,ENV2=v" or lib_exp names) -> gpurun_out/ab.log
#   py SCRIPT [ARGS]          python SCRIPT ARGS
#   trace NAME [BENCH_ARGS]   rocprofv3 --kernel-trace --stats of `bench.py --timed-only BENCH_ARGS` -> gpurun_out/trace_NAME/
#   pmc LEG [BENCH_ARGS]      LEG = solo (one frame at a time) | inflight (the default bench's timed region): four separate
#                             rocprofv3 --pmc passes (never combined with a tracing domain) of `bench.py --timed-only`
#                             -> gpurun_out/pmc/LEG_{sq1,sq2,fetch,write} (scripts/pmc_summary.py, make_blend_traffic.py)
# A failing task stops the chain: no GPU step is started behind a timeout.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run_task() {
  local task=$1; shift
  case $task in
    test)  timeout -k 10 1100 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1; local rc=$?
           tail -5 gpurun_out/pytest_gpu.log; return $rc ;;
    bench) local name=$1; shift
           timeout -k 10 600 python bench.py "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; local rc=$?
           tail -c 400 gpurun_out/bench_$name.json; echo; return $rc ;;
    ab)    timeout -k 10 1000 python scripts/ab_bench.py "$@" 2>&1 | tee -a gpurun_out/ab.log; return ${PIPESTATUS[0]} ;;
    py)    timeout -k 10 1000 python "$@" ;;
    trace) local name=$1; shift; rm -rf gpurun_out/trace_$name
           (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$name -- \
              python3 $R/bench.py --timed-only --no-cpu-baseline "$@" > $R/gpurun_out/trace_$name.log 2>&1) ;;
    pmc)   local leg=$1; shift; mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/${leg}_*
           local legargs=""; [ "$leg" = solo ] && legargs="--frames-in-flight 1"
           python -c "import sys; sys.path[:0]=['$R','$R/gsplat.js_amd/py']; import gsplat_hip as g; print(g.build_id())" > gpurun_out/pmc/build_id.txt
           local -a passes=("sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
                            "sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"
                            "fetch FETCH_SIZE" "write WRITE_SIZE")
           local p
           for p in "${passes[@]}"; do
             local pname=${p%% *} counters=${p#* }
             (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d $R/gpurun_out/pmc/${leg}_$pname -- \
                python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --timed-only $legargs "$@" > $R/gpurun_out/pmc/${leg}_$pname.log 2>&1) || return 1
             echo "pmc $leg $pname ok"
           done ;;
    *) echo "unknown task $task"; return 2 ;;
  esac
}
args=()
for a in "$@" --; do
  if [ "$a" = "--" ]; then
    if [ ${#args[@]} -gt 0 ]; then
      echo "== ${args[*]}"
      run_task "${args[@]}" || { echo "task failed: ${args[*]}"; exit 1; }
    fi
    args=()
  else args+=("$a"); fi
done
