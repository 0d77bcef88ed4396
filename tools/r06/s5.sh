set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s5; mkdir -p $OUT
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu_full.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d.get('hip_graph'))"; }
for r in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then D=$R/tools/r06/ab_old; else D=$R; fi
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 2>/dev/null | line "b32 $which r$r") >> $OUT/ab_same_box.txt
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 80 --warmup 8 --hip_graph on 2>/dev/null | line "b4graph $which r$r") >> $OUT/ab_same_box.txt
  done
done
(cd $R && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 256 --steps 10 --warmup 3 2>/dev/null | line "b256 new") >> $OUT/ab_same_box.txt
(cd $R/tools/r06/ab_old && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 256 --steps 10 --warmup 3 2>/dev/null | line "b256 old") >> $OUT/ab_same_box.txt
cat $OUT/ab_same_box.txt
