R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/final2
for i in 1 2 3; do timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -1; done > gpurun_out/final2/pytest_gpu_tail_final_tree_runs1to3.txt
python bench.py > gpurun_out/final2/bench_line_full_default_run.json 2>/dev/null
cat gpurun_out/final2/pytest_gpu_tail_final_tree_runs1to3.txt
