set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 3000 bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 > gpurun_out/prof_r06/pytest_gpu_tail.txt
du -sh gpurun_out/prof_r06
