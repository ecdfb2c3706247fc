#!/bin/bash
# build an experiment variant of the library: scripts/build_exp.sh NAME "-DFLAG ..."  -> gsplat.js_amd/lib_exp/NAME/libgsplat_hip.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../gsplat.js_amd/csrc"
out=../lib_exp/$name; mkdir -p $out
for f in gsr_api.cpp k_project.hip k_sort.hip k_bin.hip k_blend.hip k_scene.hip; do
  extra=""; [ $f = k_project.hip ] && extra="-fno-slp-vectorize"   # (as in the Makefile: FLAGS_k_project)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $extra $flags -x hip -c $f -o $out/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgsplat_hip.so $out/*.o
rm -f $out/*.o
echo built $out
