#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes raw output under gpurun_out/prof_<round>/ ; digest on the build side with tools/pmc_digest.py / pmc_sq_digest.py
# and copy the summaries into profiles/<round>/ (gpurun_out/ is scratch).
# PMC passes are separate from each other and use --kernel-trace only (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").
set -u
ROUND=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
# what the counters are collected ON: the hashes of the kernel sources of THIS snapshot (bench.py quotes a PMC figure only while
# the sources of the kernel in question are unchanged: bench.pmc_fresh) + the workload of each digest
python - > $OUT/meta_b32.json <<PY
import json, sys
sys.path.insert(0, "$R")
from sound_event_detection_dcase2017_task4_amd import build
json.dump({"sources": build.source_hashes(), "round": "$ROUND",
           "workload": "bench.py default: Cnn_9layers_FrameAvg bs=32 mixup (the metric's configuration), --steps 1 --warmup 1, SED_WGRAD_SIDE_STREAM=0"}, sys.stdout)
PY
sed 's/bs=32 mixup (the metric.s configuration)/B=256 mixup (BASELINE.json configs[1])/; s/bench.py default/bench.py --batch_size 256/' $OUT/meta_b32.json > $OUT/meta_b256.json
cd /tmp; export TMPDIR=/tmp
# Since round 4 bench.py's default workload is the metric's own configuration (bs=32); configs[1] needs --batch_size 256.
B="python $R/bench.py --no_cpu_baseline --no_extra --batch_size 256"
B32="python $R/bench.py --no_cpu_baseline --no_extra"
# Pass 1: the default command (weight gradients on the side stream).  Two kernels then run at once and each one's duration
# includes the time it waited for CU slots, so kernel-time sums no longer add up to the step; the conv_wino2 kernels (the
# roofline's dominant family) never run beside another kernel and are unaffected.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 5 --warmup 2 > $OUT/bench_line_under_rocprofv3.json 2> $OUT/stats.err
# Pass 2 and the counters: weight gradients on the MAIN stream (SED_WGRAD_SIDE_STREAM=0) -- clean per-kernel durations.
export SED_WGRAD_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o bench -- $B --steps 5 --warmup 2 > $OUT/bench_line_under_rocprofv3_main_stream_only.json 2> $OUT/stats_serial.err
$B --steps 10 --warmup 3 --by_shape > $OUT/bench_line_steps10_warmup3.json 2> $OUT/by_shape.txt
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sqA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "sqB SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "sqC SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "sqD MfmaUtil" "sqE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "sqF GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$name.err
done
unset SED_WGRAD_SIDE_STREAM
$B --steps 20 --warmup 3 > $OUT/bench_line_default_schedule.json 2> /dev/null
SED_WGRAD_SIDE_STREAM=0 $B --steps 20 --warmup 3 > $OUT/bench_line_main_stream_only.json 2> /dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT/timeline -o t -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/timeline.err
python $R/tools/timeline_overlap.py $OUT/timeline/t_kernel_trace.csv > $OUT/timeline_overlap_default_schedule.txt 2>&1
rm -rf $OUT/timeline
# the metric's own batch size (bs=32): serial kernel trace + per-step gap digest + by-shape table
SED_WGRAD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32 -o bench -- $B32 --steps 6 --warmup 3 > $OUT/bench_line_b32_under_rocprofv3_main_stream_only.json 2> $OUT/stats_b32.err
python $R/tools/step_gaps.py $(find $OUT/stats_b32 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b32_main_stream_only.txt 2>&1
python $R/tools/step_gaps.py $(find $OUT/stats_serial -name "*kernel_trace.csv") 4 > $OUT/step_digest_b256_main_stream_only.txt 2>&1
$B32 --steps 40 --warmup 5 --by_shape > $OUT/bench_line_b32.json 2> $OUT/by_shape_b32.txt
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sqD MfmaUtil" "sqB SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "sqF GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  SED_WGRAD_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_b32_$name -o p -- $B32 --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_b32_$name.err
done
# 4 clips per GPU (the strong-scaling regime: --batch_size 32 over 8 GPUs): serial kernel trace + digest, eager and graphed lines
SED_WGRAD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b4 -o bench -- $B32 --batch_size 4 --steps 6 --warmup 3 > $OUT/bench_line_b4_under_rocprofv3_main_stream_only.json 2> $OUT/stats_b4.err
python $R/tools/step_gaps.py $(find $OUT/stats_b4 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b4_main_stream_only.txt 2>&1
$B32 --batch_size 4 --steps 60 --warmup 8 --by_shape > $OUT/bench_line_b4_eager.json 2> $OUT/by_shape_b4.txt
$B32 --batch_size 4 --steps 60 --warmup 8 --hip_graph on > $OUT/bench_line_b4_hip_graph.json 2> /dev/null
# the driver's own command, complete (extra_configs, strict_fp32, cpu_baseline)
python $R/bench.py > $OUT/bench_line_full_default_run.json 2> /dev/null
find $OUT -name "*kernel_trace.csv" -size +30M -delete
ls $OUT
