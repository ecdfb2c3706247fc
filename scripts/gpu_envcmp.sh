#!/bin/bash
# bench.py under two settings of one environment variable, interleaved: VAR=GSR_ITEM_ORDER A=bin B=layer [CFGS="C3 C2"] bash scripts/gpu_envcmp.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for cfg in ${CFGS:-C3}; do for rep in 1 2 3; do for val in $A $B; do for f in 1 3; do
  st=480; [ $cfg = C4 ] && st=60
  v=$(env $VAR=$val timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight $f --steps $st --warmup 30 --timing-interval 100000 --config $cfg 2>>gpurun_out/envcmp.err | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "$cfg rep $rep $VAR=$val inflight $f fps $v"
done; done; done; done
