set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s11; mkdir -p $OUT
for b in 4 8; do for ks in 0 1 2 4 8; do
  SED_CONV_SMALLM_KS=$ks timeout 300 python tools/tail_split_bench.py --batch $b --reps 50 2>/dev/null | grep "125x8\|250x16\|TOTAL" | sed "s/^/b$b ks$ks: /" >> $OUT/ks_sweep.txt
done; done
cat $OUT/ks_sweep.txt
