set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests/test_gpu_sf16.py -x -q -m gpu -k "split or tail or xcd" 2>&1 | tail -15 > gpurun_out/s1/t_sf16_new.txt
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_optim.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s1/t_cli_optim.txt
for m in 0 2 3 4 1; do SED_CONV_TAIL=$m SED_CONV_TAIL_MINK=2 timeout 300 python tools/tail_split_bench.py --batch 32 > gpurun_out/s1/tail_mode${m}_mink2.txt 2>&1; done
for m in 0 1; do for r in 1 2; do SED_CONV_TAIL=$m timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 2>/dev/null | tail -1 > gpurun_out/s1/bench_b32_tail${m}_r$r.json; done; done
SED_CONV_TAIL=0 timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 40 --warmup 5 --by_shape 2> gpurun_out/s1/by_shape_tail0.txt >/dev/null
SED_CONV_TAIL=1 timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 40 --warmup 5 --by_shape 2> gpurun_out/s1/by_shape_tail1.txt >/dev/null
ls gpurun_out/s1
