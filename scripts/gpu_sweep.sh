#!/bin/bash
# sweep of frames in flight x compositor grid x item budget with bench.py (timed leg only)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for F in 2 3 4; do for G in 1536 1792; do for T in 1300 2500 5000; do
  v=$(GSR_BLEND_GRID=$G GSR_SEG_TARGET=$T timeout -k 10 120 python bench.py --timed-only --steps 360 --warmup 30 --frames-in-flight $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "F=$F grid=$G seg_target=$T fps=$v"
done; done; done | tee gpurun_out/sweep.txt
