set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "flip_free" -s 2>&1 | grep -E "FLIPFREE_REPORT|passed|failed|Error|assert" > gpurun_out/s2/t_flipfree.txt
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_optim.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/s2/t_cli_optim.txt
timeout 1500 python -m pytest tests/test_gpu_parallel.py -x -q -m gpu -k "eight" 2>&1 | tail -25 > gpurun_out/s2/t_eight.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s2/bench_default.json 2> gpurun_out/s2/bench_default.err
ls gpurun_out/s2
