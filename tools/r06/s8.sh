set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s8; mkdir -p $OUT
timeout 600 python tools/wgrad_reduce_bench.py > $OUT/wgrad_reduce_bench.txt 2>&1
cat $OUT/wgrad_reduce_bench.txt
