#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for F in 3; do for G in 1024 1280 1536; do for T in 500 800 1300; do
  v=$(GSR_BLEND_GRID=$G GSR_SEG_TARGET=$T timeout -k 10 120 python bench.py --timed-only --steps 360 --warmup 30 --frames-in-flight $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "F=$F grid=$G seg_target=$T fps=$v"
done; done; done | tee gpurun_out/sweep2.txt
