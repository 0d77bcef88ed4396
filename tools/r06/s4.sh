set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s4; mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu_full.txt
for r in 1 2; do timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 2>/dev/null | tail -1 > $OUT/bench_b32_r$r.json; done
for r in 1 2; do timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 60 --warmup 8 --hip_graph on 2>/dev/null | tail -1 > $OUT/bench_b4_graph_r$r.json; done
timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 40 --warmup 5 --by_shape 2> $OUT/by_shape_b32.txt > /dev/null
cd /tmp; export TMPDIR=/tmp
SED_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b4 -o bench -- python $R/bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 6 --warmup 3 > $OUT/bench_b4_under_rocprof.json 2> $OUT/stats_b4.err
python $R/tools/step_gaps.py $(find $OUT/stats_b4 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b4_main_stream_only.txt 2>&1
SED_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32 -o bench -- python $R/bench.py --no_cpu_baseline --no_extra --steps 6 --warmup 3 > $OUT/bench_b32_under_rocprof.json 2> $OUT/stats_b32.err
python $R/tools/step_gaps.py $(find $OUT/stats_b32 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b32_main_stream_only.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.csv" -size +2M -delete
ls $OUT
