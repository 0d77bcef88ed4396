set -u
R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SED_WGRAD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b4 -o bench -- python $R/bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 6 --warmup 3 > $OUT/bench_b4_under_rocprof.json 2> $OUT/stats_b4.err
python $R/tools/step_gaps.py $(find $OUT/stats_b4 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b4_main_stream_only.txt 2>&1
python $R/bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 40 --warmup 5 --by_shape > $OUT/bench_b4.json 2> $OUT/by_shape_b4.txt
python $R/bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 40 --warmup 5 --hip_graph on > $OUT/bench_b4_graph.json 2> /dev/null
python $R/bench.py --no_cpu_baseline --no_extra --steps 40 --warmup 5 --by_shape > $OUT/bench_b32.json 2> $OUT/by_shape_b32.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
