#!/bin/bash
# CPU sanitizer run of the oracle (ASan + UBSan): builds an instrumented liboracle.so, runs the oracle / scene tests
# against it, restores the normal build.  (GPU sanitizers are not available on the pool; this covers the CPU code.)
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -ffp-contract=off -mfma -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -shared -o /tmp/liboracle_asan.so oracle/oracle.c -lm
cp oracle/liboracle.so /tmp/liboracle_orig.so
trap 'cp /tmp/liboracle_orig.so oracle/liboracle.so' EXIT
cp /tmp/liboracle_asan.so oracle/liboracle.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle_sort.py tests/test_oracle_render.py tests/test_scene_camera.py tests/test_shader_golden.py tests/test_host_golden.py -x -q -m "not gpu"
