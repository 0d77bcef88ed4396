#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes raw output under gpurun_out/prof_<round>/ ; digest on the build side with tools/pmc_digest.py / pmc_sq_digest.py
# and copy the summaries into profiles/<round>/ (gpurun_out/ is scratch).
# PMC passes are separate from each other and use --kernel-trace only (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").
set -u
ROUND=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 5 --warmup 2 > $OUT/bench_line_under_rocprofv3.json 2> $OUT/stats.err
$B --steps 10 --warmup 3 --by_shape > $OUT/bench_line_steps10_warmup3.json 2> $OUT/by_shape.txt
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sqA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "sqB SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "sqC SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "sqD MfmaUtil"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$name.err
done
ls $OUT
