#!/bin/bash
# the driver's short run: bench.py --steps 20 with a small warm-up, several times (fresh process each)
cd $GRAFT_REPO_ROOT
for sf in 4 12 40; do for rep in 1 2 3; do
  GSR_BENCH_SETUP_FRAMES=$sf timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('setup $sf rep $rep: %.0f fps  %.4f ms' % (d['value'], d['ms_per_step']))"
done; done
