set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s3
timeout 600 python tools/flipfree_diag.py Cnn_9layers_FrameAvg > gpurun_out/s3/flipfree_diag_FrameAvg.txt 2>&1
timeout 600 python tools/flipfree_diag.py Cnn_9layers_Gru_FrameAtt > gpurun_out/s3/flipfree_diag_Gru.txt 2>&1
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s3/pytest_gpu_full.txt
ls gpurun_out/s3
