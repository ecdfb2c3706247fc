#!/bin/bash
# copy the round's evidence from gpurun_out/ (scratch) into profiles/ (tracked): run after scripts/gpu_r2_profiles.sh
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/r02
cp $R/bench_c3.json profiles/r02_bench_c3_exact.json
cp $R/bench_c3_earlyout.json profiles/r02_bench_c3_earlyout.json
cp $R/bench_c2.json profiles/r02_bench_c2_exact.json
cp $R/bench_c4.json profiles/r02_bench_c4_exact.json
cp $(ls -t $R/trace_1inflight/*/*kernel_stats.csv | head -1) profiles/r02_kernel_stats_c3_exact_1inflight.csv
cp $(ls -t $R/trace_3inflight/*/*kernel_stats.csv | head -1) profiles/r02_kernel_stats_c3_exact_3inflight.csv
python scripts/pmc_summary.py gpurun_out/pmc > profiles/r02_pmc_c3_exact.txt
python scripts/make_blend_traffic.py gpurun_out/pmc > /dev/null
for f in blend_stamps kernel_stamps valu_cost4; do [ -f gpurun_out/$f.txt ] && cp gpurun_out/$f.txt profiles/r02_$f.txt; done
[ -f gpurun_out/r02/kernel_stats_20M.csv ] && cp gpurun_out/r02/kernel_stats_20M.csv profiles/r02_kernel_stats_20M_1080p.csv
ls profiles
