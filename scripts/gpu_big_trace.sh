#!/bin/bash
# kernel traces of big_scene_frames.py for several library builds: LIBS="base xcd" [N=20000000] bash scripts/gpu_big_trace.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
for LIB in $LIBS; do
  rm -rf gpurun_out/bigtrace_$LIB
  if [ "$LIB" != "base" ]; then export GSPLAT_HIP_LIB=$GRAFT_REPO_ROOT/gsplat.js_amd/lib_exp/$LIB/libgsplat_hip.so; else unset GSPLAT_HIP_LIB; fi
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bigtrace_$LIB -- python3 $GRAFT_REPO_ROOT/scripts/big_scene_frames.py ${N:-20000000} ${FRAMES:-6} ${W:-1920} ${H:-1080} > $GRAFT_REPO_ROOT/gpurun_out/bigtrace_$LIB.log 2>&1) || { tail -5 gpurun_out/bigtrace_$LIB.log; exit 1; }
  echo "== $LIB"; grep "^frame" gpurun_out/bigtrace_$LIB.log | tail -2
  python3 - <<PY
import csv,glob
fs=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/bigtrace_$LIB/*/*kernel_stats.csv")
for r in csv.DictReader(open(fs[0])):
    nm=r["Name"].split("(")[0].replace("void ","").replace("gsr::","")
    if nm.startswith("k_") and not nm.startswith("k_build") and not nm.startswith("k_clear"):
        print("  %-30s calls %4s avg %8.1f min %8.1f us" % (nm[:30], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
