#!/bin/bash
# Same-box A/B of the current tree against an earlier commit (boxes of the pool differ by +-2-3 %, so numbers from different
# leases cannot be compared).  In the build container:
#     git worktree add tools/ab_old <commit> && (cd tools/ab_old && python -m sound_event_detection_dcase2017_task4_amd.build)
# (tools/ab_old/ is git-ignored; its built .so travels to the GPU box with the snapshot), then on the GPU box:
#     gpurun --timeout 1800 -- 'bash tools/ab_same_box.sh'
# Alternating runs of both trees: the metric's batch (event-free region), 4 / 8 clips per GPU under the HIP graph, 16 clips, B=256.
# Round 6's result: profiles/r06/README.md.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/ab; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d.get('hip_graph'))"; }
for r in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then D=$R/tools/ab_old; else D=$R; fi
    (cd $D && python bench.py --no_cpu_baseline --no_extra --steps 60 --warmup 8 2>/dev/null | line "b32 $which r$r") >> $OUT/ab_same_box.txt
    (cd $D && python bench.py --no_cpu_baseline --no_extra --batch_size 4 --steps 80 --warmup 8 --hip_graph on 2>/dev/null | line "b4graph $which r$r") >> $OUT/ab_same_box.txt
    (cd $D && python bench.py --no_cpu_baseline --no_extra --batch_size 8 --steps 80 --warmup 8 --hip_graph on 2>/dev/null | line "b8graph $which r$r") >> $OUT/ab_same_box.txt
    (cd $D && python bench.py --no_cpu_baseline --no_extra --batch_size 16 --steps 60 --warmup 8 2>/dev/null | line "b16 $which r$r") >> $OUT/ab_same_box.txt
  done
done
for which in old new; do
  if [ $which = old ]; then D=$R/tools/ab_old; else D=$R; fi
  (cd $D && python bench.py --no_cpu_baseline --no_extra --batch_size 256 --steps 10 --warmup 3 2>/dev/null | line "b256 $which") >> $OUT/ab_same_box.txt
done
cat $OUT/ab_same_box.txt
