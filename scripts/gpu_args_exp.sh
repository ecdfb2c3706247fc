#!/bin/bash
# bench under a list of argument sets: CASES="--a 1|--b 2 --c" ('|' separates cases)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/args_*
i=0
IFS='|' read -ra cases <<< "$CASES"
for c in "${cases[@]}"; do
  i=$((i+1))
  timeout -k 10 200 python bench.py --steps 240 --warmup 20 --no-cpu-baseline $c > gpurun_out/args_${i}.json 2> gpurun_out/args_${i}.err; rc=$?
  python - "$c" gpurun_out/args_${i}.json $rc <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print("%-56s fps %7.1f ms %.3f"%(sys.argv[1], d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1],"rc",sys.argv[3],"ERR",e)
PY
done
