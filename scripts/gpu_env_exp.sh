#!/bin/bash
# bench under a list of environment settings: ENVS="A=1;B=2 C=3" (space separated cases, ; inside a case)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/env_*
i=0
for case in base $ENVS; do
  i=$((i+1))
  ( if [ "$case" != base ]; then IFS=';' read -ra kv <<< "$case"; for x in "${kv[@]}"; do export "$x"; done; fi
    timeout -k 10 200 python bench.py --steps 120 --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/env_${i}.json 2> gpurun_out/env_${i}.err; echo "$case rc=$?"
    python - "$case" gpurun_out/env_${i}.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print("%-44s fps %7.1f"%(sys.argv[1], d["value"]), {k:round(v,4) for k,v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1],"ERR",e)
PY
  )
done
