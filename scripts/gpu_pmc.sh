#!/bin/bash
# PMC passes over the bench (separate rocprofv3 runs; --pmc is never combined with tracing domains).
cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE
