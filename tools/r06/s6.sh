set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/s6; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_sf16.py tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_subset.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('hip_graph'))"; }
for r in 1 2; do
  for which in old new; do
    if [ $which = old ]; then D=$R/tools/r06/ab_old; else D=$R; fi
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 80 --warmup 8 --hip_graph on 2>/dev/null | line "b4graph $which r$r") >> $OUT/ab.txt
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 8 --steps 80 --warmup 8 --hip_graph on 2>/dev/null | line "b8graph $which r$r") >> $OUT/ab.txt
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 16 --steps 60 --warmup 8 2>/dev/null | line "b16 $which r$r") >> $OUT/ab.txt
    (cd $D && timeout 300 python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 --no_kernel_events 2>/dev/null | line "b32 $which r$r") >> $OUT/ab.txt
  done
done
(cd $R && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 40 --warmup 5 --by_shape 2> $OUT/by_shape_b4_new.txt >/dev/null)
(cd $R/tools/r06/ab_old && timeout 300 python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 40 --warmup 5 --by_shape 2> $OUT/by_shape_b4_old.txt >/dev/null)
cat $OUT/ab.txt
