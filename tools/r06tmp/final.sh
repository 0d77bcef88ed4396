set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 3000 bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -1; done > gpurun_out/prof_r06/pytest_gpu_tail_runs1to3.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/prof_r06/smoke.txt 2>&1
cat gpurun_out/prof_r06/pytest_gpu_tail_runs1to3.txt gpurun_out/prof_r06/smoke.txt
