#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do for t in 1100 1200 1300 1400 1500 1650; do
  v=$(GSR_SEG_TARGET=$t timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight 3 --steps 480 --warmup 30 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "rep $rep seg_target $t fps $v"
done; done
