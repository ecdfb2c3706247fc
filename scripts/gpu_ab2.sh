#!/bin/bash
# A/B in one process: VARIANTS="base pk" [ARGS="--config C3"] [TESTLIB=pk] bash scripts/gpu_ab2.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
if [ -n "$TESTLIB" ]; then
  GSPLAT_HIP_LIB=$GRAFT_REPO_ROOT/gsplat.js_amd/lib_exp/$TESTLIB/libgsplat_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q $TESTARGS > gpurun_out/pytest_$TESTLIB.log 2>&1; rc=$?
  echo "pytest $TESTLIB rc=$rc"; tail -4 gpurun_out/pytest_$TESTLIB.log
  [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 600 python scripts/ab_bench.py --check $ARGS $VARIANTS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_$(echo $VARIANTS | tr ' ' '_').txt
if [ -n "$ARGS2" ]; then timeout -k 10 600 python scripts/ab_bench.py $ARGS2 $VARIANTS 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_$(echo $VARIANTS | tr ' ' '_').txt; fi
