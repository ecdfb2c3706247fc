#!/bin/bash
# The round's bench lines beside the default one: C4 as the timed workload, the per-rank ceilings of a 2/4/8-GPU run
# (bench.py --emulate-rank), the JavaScript host.  Run on the GPU box: bash scripts/final_lines.sh  -> gpurun_out/lines_*.json
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python bench.py --config C4 --steps 120 --warmup 12 --no-cpu-baseline --no-other-configs > gpurun_out/lines_c4.json 2> gpurun_out/lines_c4.err || exit 1
python bench.py --config C2 --steps 480 --warmup 30 --no-cpu-baseline --no-other-configs > gpurun_out/lines_c2.json 2> gpurun_out/lines_c2.err || exit 1
: > gpurun_out/lines_ranks.txt
for cfg in C3 C4; do for q in 0/1 0/2 1/4 3/8; do
  st=240; [ $cfg = C4 ] && st=60
  if [ $q = 0/1 ]; then emu=""; else emu="--emulate-rank $q"; fi
  v=$(python bench.py --no-cpu-baseline --timed-only --steps $st --warmup 20 --config $cfg $emu 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "$cfg rank $q frames_per_sec $v" >> gpurun_out/lines_ranks.txt
done; done
cat gpurun_out/lines_ranks.txt
if command -v node > /dev/null; then
  python -c "import sys; sys.path[:0]=['.','gsplat.js_amd/py']; import gsplat_hip as g; g.synth.config_rows('C3').tofile('/tmp/c3.splat')"
  node tools/bench_node.js /tmp/c3.splat 1920 1080 1132 240 20 > gpurun_out/lines_node_c3.json 2> gpurun_out/lines_node_c3.err || exit 1
  tail -c 300 gpurun_out/lines_node_c3.json
fi
