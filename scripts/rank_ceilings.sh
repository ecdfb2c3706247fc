cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
: > gpurun_out/lines_ranks.txt
for cfg in C3 C4; do for q in 0/1 0/2 1/4 3/8; do
  st=240; [ $cfg = C4 ] && st=60
  if [ $q = 0/1 ]; then emu=""; else emu="--emulate-rank $q"; fi
  v=$(python bench.py --no-cpu-baseline --timed-only --steps $st --warmup 20 --config $cfg $emu 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f %s" % (d["value"], json.dumps(d["stage_ms"])))') || exit 1
  echo "$cfg rank $q frames_per_sec $v" >> gpurun_out/lines_ranks.txt
done; done
cat gpurun_out/lines_ranks.txt
