#!/bin/bash
# build an experiment variant of the library: scripts/build_exp.sh NAME "-DFLAG ..."  -> gsplat.js_amd/lib_exp/NAME/libgsplat_hip.so
# Same sources, flags and per-file options as the shipped build: it IS the csrc Makefile, with another output directory and
# the extra defines (a failing compile fails the script: make, not background jobs and a bare `wait`).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../gsplat.js_amd/csrc"
make --no-print-directory -j6 OUT=../lib_exp/$name EXTRA="$flags" lib
echo built ../lib_exp/$name
