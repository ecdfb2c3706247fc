#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for g in 768 1024 1280 1536 1792; do
  v=$(GSR_BLEND_GRID=$g timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight ${F:-3} --steps 480 --warmup 30 ${CFG:-} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "rep $rep grid $g fps $v"
done; done
