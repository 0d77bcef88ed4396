set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s9; mkdir -p $OUT
for m in 512 100000; do
  SED_CONV_SMALL_MAX=$m timeout 300 python tools/tail_split_bench.py --batch 32 > $OUT/layers_b32_smallmax$m.txt 2>&1
  SED_CONV_SMALL_MAX=$m timeout 300 python tools/tail_split_bench.py --batch 256 --reps 8 > $OUT/layers_b256_smallmax$m.txt 2>&1
  for r in 1 2; do SED_CONV_SMALL_MAX=$m timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 --no_kernel_events 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b32 smallmax $m', d['ms_per_step'])" >> $OUT/ab.txt; done
done
cat $OUT/ab.txt
