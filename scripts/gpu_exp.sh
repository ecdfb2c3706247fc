#!/bin/bash
# run bench stage timings for each experiment library under gsplat.js_amd/lib_exp/ (and env sweeps in $SWEEP)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/exp_*
for d in base $(ls gsplat.js_amd/lib_exp 2>/dev/null); do
  if [ $d = base ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$GRAFT_REPO_ROOT/gsplat.js_amd/lib_exp/$d/libgsplat_hip.so; fi
  timeout -k 10 200 python bench.py --steps 120 --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/exp_$d.json 2> gpurun_out/exp_$d.err; echo "$d rc=$?"
done
unset GSPLAT_HIP_LIB
for g in $GRIDS; do
  GSR_BLEND_GRID=$g timeout -k 10 200 python bench.py --steps 120 --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/exp_grid$g.json 2> gpurun_out/exp_grid$g.err; echo "grid$g rc=$?"
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/exp_*.json")):
    try:
        d=json.load(open(f)); print("%-28s fps %7.1f"%(os.path.basename(f), d["value"]), {k:round(v,4) for k,v in d["stage_ms"].items()})
    except Exception as e: print(f,"ERR",e)
PY
