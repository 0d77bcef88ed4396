#!/bin/bash
# PMC counters of the split-f16 kernels on one layer (run through gpurun from the repo root):
#   bash tools/pmc_conv.sh <layer index into tools/conv_bench.py LAYERS> <only: sf16|sf16w> <tag>
# separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md)
set -u
L=${1:-6}; ONLY=${2:-sf16}; TAG=${3:-pmc_conv}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/conv_bench.py --batch 128 --reps 2 --only $ONLY --layers $L"
for pass in "a SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "b SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "c SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
            "e SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "f GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o p -- $CMD > $O/$name.out 2> $O/$name.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$O/*/p_counter_collection.csv"):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        if "sf16" not in k or "reduce" in k or "pack" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[(k, r["Counter_Name"])] += 1
    for (k, c), n in seen.items(): cnt[(k, c)] = n
out = open("$O/summary.txt", "w")
for k, d in sorted(agg.items()):
    out.write(k + "\n")
    for c, v in sorted(d.items()):
        out.write("   %-28s %16.0f  per dispatch %14.0f (%d dispatches)\n" % (c, v, v / max(cnt[(k, c)], 1), cnt[(k, c)]))
    g = lambda n: d.get(n, 0.0)
    if g("SQ_INSTS_MFMA"):
        out.write("   derived: VALU(non-MFMA)/MFMA %.2f  SALU/MFMA %.2f  LDS/MFMA %.2f\n" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA") if g("SQ_INSTS_VALU") > g("SQ_INSTS_MFMA") else g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_SALU") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA")))
    if g("SQ_BUSY_CYCLES"):
        out.write("   derived: MFMA busy / SQ busy (per-SE aggregates) %.3f\n" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
    if g("SQ_WAVE_CYCLES"):
        out.write("   derived: of wave cycles: WAIT_ANY %.3f WAIT_INST_ANY %.3f ACTIVE_INST_ANY %.3f WAIT_INST_LDS %.3f\n" % tuple(g(n) / g("SQ_WAVE_CYCLES") for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")))
    if g("SQ_LDS_IDX_ACTIVE"):
        out.write("   derived: LDS bank conflict cycles / LDS active cycles %.3f\n" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
out.close()
PY
cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete
