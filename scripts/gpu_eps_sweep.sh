#!/bin/bash
# how much of the compositor's work lies behind exact saturation (T == 0)?  early-out thresholds down to the denormal floor
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for e in 1e-4 1e-10 1e-20 1e-30 1e-37 1e-44; do
  timeout -k 10 300 python bench.py --steps 120 --warmup 10 --early-out-eps $e --no-cpu-baseline > gpurun_out/eps_$e.json 2> gpurun_out/eps_$e.err || { tail -3 gpurun_out/eps_$e.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/eps_$e.json").read().strip().splitlines()[-1])
o=d.get("one_frame_in_flight",{})
print("eps $e fps %.0f solo %.0f stage %s" % (d["value"], o.get("frames_per_sec",0), {k: round(v*1e3,1) for k,v in o.get("stage_ms",{}).items()}))
PY
done
