#!/bin/bash
# PMC passes over the bench (separate rocprofv3 runs; --pmc is never combined with tracing domains).  Two legs, each its own
# set of runs so that every k_blend dispatch of a run belongs to ONE configuration: `inflight` = the timed region of the
# default bench (three throughput contexts), `solo` = one frame at a time on an exact context.
cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
run() {
  leg=$1; name=$2; shift; shift
  if [ $leg = solo ]; then legargs="--frames-in-flight 1"; else legargs=""; fi
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${leg}_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --no-cpu-baseline --timed-only $legargs $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc/${leg}_$name.log 2>&1
  echo "$leg $name rc=$?"
}
for leg in inflight solo; do
run $leg sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY && \
run $leg sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE && \
run $leg fetch FETCH_SIZE && \
run $leg write WRITE_SIZE || exit 1
done
