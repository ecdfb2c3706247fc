#!/bin/bash
# kernel trace of one library build, one frame in flight: LIB=fold2 [ARGS="--config C3"] bash scripts/gpu_trace_lib.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -rf gpurun_out/trace_$LIB
export TMPDIR=/tmp
if [ "$LIB" != "base" ]; then export GSPLAT_HIP_LIB=$GRAFT_REPO_ROOT/gsplat.js_amd/lib_exp/$LIB/libgsplat_hip.so; fi
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$LIB -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --timed-only --frames-in-flight 1 --steps 120 --warmup 10 $ARGS > $GRAFT_REPO_ROOT/gpurun_out/trace_$LIB.log 2>&1; echo "prof rc=$?"
python3 - <<PY
import csv,glob
fs=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace_$LIB/*/*kernel_stats.csv")
for r in csv.DictReader(open(fs[0])):
    print("  %-30s calls %4s avg %8.1f us %6s%%" % (r["Name"].split("(")[0].replace("void ","").replace("gsr::","")[:30], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
