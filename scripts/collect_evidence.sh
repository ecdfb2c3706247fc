#!/bin/bash
# Copies what the GPU runs left under gpurun_out/ (scripts/gpu.sh pmc / trace, blend_stamps.py, kernel_stamps.py, policy_check.py,
# bench.py, final_lines.sh, rank_ceilings.sh, big_scene_check.py, pytest) into profiles/r04_* and rebuilds blend_traffic.json /
# r04_pmc_c3.txt.  Run in the build container after the calls:  bash scripts/collect_evidence.sh [batch1|batch2]
cd "$(dirname "$0")/.."
id=$(cat gpurun_out/pmc/build_id.txt)
if [ "$1" != batch2 ]; then
  python scripts/make_blend_traffic.py > /dev/null
  (echo "# rocprofv3 --pmc passes (scripts/gpu.sh pmc solo / pmc inflight), build $id: median per kernel; one counter set per run, never combined with tracing"
   echo "## one frame at a time (bench.py --timed-only --frames-in-flight 1)"; python scripts/pmc_summary.py gpurun_out/pmc solo; echo
   echo "## the timed region of the default bench (three frames in flight, throughput contexts; kernels serialised by the profiler)"; python scripts/pmc_summary.py gpurun_out/pmc inflight) > profiles/r04_pmc_c3.txt
  for n in c3_3f:c3_3inflight c3_1f:c3_1inflight c3_1f_tp:c3_1inflight_throughput c4_1f:c4_1inflight c4_3f:c4_3inflight; do
    f=$(ls -t gpurun_out/trace_${n%%:*}/*/*kernel_stats.csv | head -1); cp "$f" profiles/r04_kernel_stats_${n##*:}.csv
  done
  (echo "pytest tests -m gpu on the box, build $id:"; tail -3 gpurun_out/pytest_gpu.log) > profiles/r04_pytest_gpu.txt
fi
if [ "$1" != batch1 ]; then
  cp gpurun_out/bench_default.json profiles/r04_bench_c3.json; cp gpurun_out/lines_c4.json profiles/r04_bench_c4.json
  cp gpurun_out/lines_c2.json profiles/r04_bench_c2.json; cp gpurun_out/lines_node_c3.json profiles/r04_bench_node_c3.json
  grep -v amdgpu.ids gpurun_out/blend_stamps.txt > profiles/r04_blend_stamps.txt; grep -v amdgpu.ids gpurun_out/kernel_stamps.txt > profiles/r04_kernel_stamps.txt
  grep -v amdgpu.ids gpurun_out/policy_check.txt > profiles/r04_policy_check.txt; cp gpurun_out/lines_ranks.txt profiles/r04_rank_ceilings.txt
  (echo "scripts/big_scene_check.py at 20 M splats / 1080p and 12 M / 4K, build $id (first frames of fresh contexts: the times include one-time work)"; grep -v amdgpu.ids gpurun_out/big_scene_check.txt) > profiles/r04_big_scene_check.txt
fi
echo "collected for build $id"
