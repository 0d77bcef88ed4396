R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/flaky2
for i in $(seq 1 14); do timeout 900 python -m pytest tests/test_gpu_parallel.py -q -m gpu -x -k "one_rank_over_rccl" 2>&1 | tail -40 > gpurun_out/flaky2/run$i.txt; tail -1 gpurun_out/flaky2/run$i.txt; done
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_cli.py -q -m gpu -x 2>&1 | tail -2
