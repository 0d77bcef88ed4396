#!/bin/bash
# Digest gpurun_out/prof_<round>/ (written on the GPU box by tools/collect_profiles.sh) into profiles/<round>/ (tracked).
#   bash tools/store_profiles.sh r04 [gpurun_out/<dir with bench_full_default.json and pytest_gpu.log>]
set -eu
ROUND=${1:-r04}
EXTRA=${2:-}
P=profiles/$ROUND; G=gpurun_out/prof_$ROUND
mkdir -p $P
python tools/pmc_digest.py $G/pmc_fetch $G/pmc_write $G/meta_b256.json > $P/pmc_traffic.json
python tools/pmc_digest.py $G/pmc_b32_fetch $G/pmc_b32_write $G/meta_b32.json > $P/pmc_traffic_b32.json
python tools/pmc_mfma_busy.py $G/pmc_sqD "B=256 training step (configs[1])" $G/meta_b256.json > $P/pmc_mfma_busy.json
python tools/pmc_mfma_busy.py $G/pmc_b32_sqD "bs=32 training step (the metric's configuration)" $G/meta_b32.json > $P/pmc_mfma_busy_b32.json
python tools/pmc_sq_digest.py $G/pmc_b32_sqD $G/pmc_b32_sqB $G/pmc_b32_sqF > $P/rocprofv3_pmc_summaries_b32.txt 2>&1
python tools/pmc_sq_digest.py $G/pmc_sqA $G/pmc_sqB $G/pmc_sqC $G/pmc_sqD $G/pmc_sqE $G/pmc_sqF > $P/rocprofv3_pmc_summaries.txt 2>&1
cp $G/stats/bench_kernel_stats.csv $P/rocprofv3_kernel_stats__b256_steps5_warmup2.csv
cp $G/stats_serial/bench_kernel_stats.csv $P/rocprofv3_kernel_stats__main_stream_only__b256_steps5_warmup2.csv
cp $G/stats_b32/bench_kernel_stats.csv $P/rocprofv3_kernel_stats__main_stream_only__b32_steps6_warmup3.csv
for f in bench_line_under_rocprofv3.json bench_line_under_rocprofv3_main_stream_only.json bench_line_b32_under_rocprofv3_main_stream_only.json \
         bench_line_default_schedule.json bench_line_main_stream_only.json bench_line_steps10_warmup3.json bench_line_b32.json \
         step_digest_b256_main_stream_only.txt step_digest_b32_main_stream_only.txt timeline_overlap_default_schedule.txt; do
  cp $G/$f $P/$f
done
cp $G/by_shape.txt $P/by_shape__b256_steps10_warmup3.txt
cp $G/by_shape_b32.txt $P/by_shape__b32_steps40_warmup5.txt
for f in step_digest_b4_main_stream_only.txt bench_line_b4_eager.json bench_line_b4_hip_graph.json bench_line_full_default_run.json; do
  if [ -f $G/$f ]; then cp $G/$f $P/$f; fi
done
if [ -f $G/by_shape_b4.txt ]; then cp $G/by_shape_b4.txt $P/by_shape__b4_steps60_warmup8.txt; fi
if [ -n "$EXTRA" ]; then
  cp $EXTRA/bench_full_default.json $P/bench_line_full_default_run.json
  tail -3 $EXTRA/pytest_gpu.log > $P/pytest_gpu_tail.txt
fi
