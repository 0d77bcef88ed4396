#!/bin/bash
# PMC passes over the Winograd experiment kernel (one layer, dgrad form): bash tools/experiments/conv_wsf16/pmc.sh <outdir> [layer index]
set -u
R=$PWD; OUT=$R/$1; L=${2:-5}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "a MfmaUtil" "b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
            "d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "e SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "f GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python $R/tools/experiments/conv_wsf16/run.py bench --batch 32 --layers $L --only_wino --reps 2 > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/pmc_sq_digest.py $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d $OUT/pmc_e $OUT/pmc_f > $OUT/summary.txt 2>&1
