#!/bin/bash
# segment-target sweep through bench.py, long-item policy off (the target alone sets the item length)
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-C2 C3 C4}; do for t in ${TARGETS:-5000 1300 650 400 200 100 50}; do for f in 1 3; do
  st=360; [ $cfg = C4 ] && st=90
  v=$(GSR_LONG_ITEMS=0 GSR_SEG_TARGET=$t timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight $f --steps $st --warmup 30 --timing-interval 100000 --config $cfg 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "$cfg target $t inflight $f fps $v"
done; done; done
