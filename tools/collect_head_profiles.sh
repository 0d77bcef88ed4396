#!/bin/bash
# Profiles of the recurrent / attention heads (BASELINE.json configs[3]: Cnn_9layers_Gru_FrameAtt; SURVEY.md 8(f)4: Transformer
# heads) at B=256, one stream: step digest + per-dispatch list of the head's kernels, and the stand-alone timings of the
# recurrences, the attention core and the dense GEMMs.   bash tools/collect_head_profiles.sh r04     (GPU box, inside gpurun)
set -u
ROUND=${1:-r04}
R=$PWD; O=$R/gpurun_out/heads_$ROUND; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in Cnn_9layers_Gru_FrameAtt Cnn_9layers_Transformer_FrameAvg; do
  rm -rf /tmp/prof_$m
  SED_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$m -o g -- python $R/bench.py --model_type $m --batch_size 256 --steps 5 --warmup 2 --no_extra --no_cpu_baseline --no_kernel_events > $O/bench_line_${m}_under_rocprofv3_main_stream_only.json 2>/dev/null
  f=$(find /tmp/prof_$m -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_gaps.py $f 4 > $O/step_digest_${m}_b256_main_stream_only.txt 2>&1
  python $R/tools/trace_pick.py $f gemm gru mha att_ head_ fill copy reduce_rows elementwise amax igemm > $O/head_dispatches_${m}_b256.txt 2>&1
done
cd $R
python tools/gru_time.py 256 32 > $O/gru_recurrence_times.txt 2>&1
python tools/mha_time.py 256 125 > $O/attention_core_times.txt 2>&1
python tools/gemm_time.py > $O/dense_gemm_times.txt 2>&1
for m in Cnn_9layers_FrameAtt Cnn_9layers_Gru_FrameAtt Cnn_9layers_FrameAvg Cnn_9layers_Transformer_FrameAvg Cnn_9layers_Transformer_FrameAtt; do
  python bench.py --model_type $m --batch_size 256 --steps 15 --warmup 4 --no_extra --no_cpu_baseline --no_kernel_events 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %8.1f clips/s %7.3f ms/step' % (d['config'].get('workload', '')[:40], d['value'], d['ms_per_step']))"
done > $O/model_step_times_b256_same_box.txt 2>&1
