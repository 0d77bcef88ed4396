set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B4="python $R/bench.py --no_cpu_baseline --no_extra --batch_size 4 --no_kernel_events"
export SED_WGRAD_SIDE_STREAM=0
for pass in "sqA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "sqB SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "sqC SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "sqD MfmaUtil" "sqE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "sqF GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $B4 --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/pmc_sq_digest.py $OUT/pmc_sqA $OUT/pmc_sqB $OUT/pmc_sqC $OUT/pmc_sqD $OUT/pmc_sqE $OUT/pmc_sqF > $OUT/pmc_b4_summaries.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b4 -o bench -- $B4 --steps 6 --warmup 3 > /dev/null 2> $OUT/stats.err
python $R/tools/step_gaps.py $(find $OUT/stats_b4 -name "*kernel_trace.csv") 4 > $OUT/step_digest_b4.txt 2>&1
find $OUT -name "*.csv" -delete
ls $OUT
