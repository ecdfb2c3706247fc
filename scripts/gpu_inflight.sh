#!/bin/bash
# frames-in-flight x compositor grid sweep through bench.py (timed region only)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
for f in 3 4 5 6; do for g in 1536 1792; do
  v=$(GSR_BLEND_GRID=$g timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight $f --steps 480 --warmup 30 ${CFG:-} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "rep $rep inflight $f grid $g fps $v"
done; done; done
