#!/bin/bash
# rehearsal of bench.py's N>1 path on a one-GPU box: WORLD ranks share GPU 0, gloo backend (slab all-gather through host)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
W=${WORLD:-2}
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $W --steps 60 --warmup 10 --backend gloo $BENCH_ARGS > gpurun_out/rehearse_$W.json 2> gpurun_out/rehearse_$W.err; echo "rehearse world=$W rc=$?"
tail -c 1500 gpurun_out/rehearse_$W.json; tail -5 gpurun_out/rehearse_$W.err
