set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s14; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_cli.py tests/test_gpu_optim.py tests/test_gpu_parallel.py -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_subset.txt
timeout 300 python tools/wgrad_reduce_bench.py 2>/dev/null | tail -16 > $OUT/wgrad_reduce_bench.txt
timeout 600 python tools/sustained_probe.py 200 > $OUT/sustained_probe.txt 2>&1
cat $OUT/pytest_subset.txt; tail -5 $OUT/sustained_probe.txt; cat $OUT/wgrad_reduce_bench.txt
