#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for cfg in C2 C4 C1; do for rep in 1 2; do for g in 1536 1792 2048; do
  st=480; [ $cfg = C4 ] && st=60
  v=$(GSR_BLEND_GRID=$g timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight 3 --steps $st --warmup 20 --config $cfg 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "$cfg rep $rep grid $g fps $v"
done; done; done
for rep in 1 2; do for e in 1e-4; do for g in 1536 1792; do
  v=$(GSR_BLEND_GRID=$g timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight 3 --steps 480 --warmup 20 --early-out-eps $e 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "C3 eps $e rep $rep grid $g fps $v"
done; done; done
