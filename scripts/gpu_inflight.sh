#!/bin/bash
# frames-in-flight sweep through bench.py (timed region only): CFG="--config C3" bash scripts/gpu_inflight.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
for f in 2 3 4 5 6; do
  v=$(timeout -k 10 120 python bench.py --no-cpu-baseline --timed-only --frames-in-flight $f --steps ${STEPS:-480} --warmup 30 ${CFG:-} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f" % d["value"])') || exit 1
  echo "rep $rep inflight $f fps $v"
done; done
