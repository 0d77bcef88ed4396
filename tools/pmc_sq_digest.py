"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs (any counters), one row per kernel and pass.
    python tools/pmc_sq_digest.py gpurun_out/pmc_sqA gpurun_out/pmc_sqB ... > profiles/r01/rocprofv3_pmc_summaries.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    per_kernel = defaultdict(dict)
    launches = defaultdict(int)
    for d in sys.argv[1:]:
        seen = defaultdict(set)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                c = r["Counter_Name"]
                per_kernel[k][c] = per_kernel[k].get(c, 0.0) + float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
        for k, s in seen.items():
            launches[k] = max(launches[k], len(s))
    counters = sorted({c for v in per_kernel.values() for c in v})
    keys = sorted(per_kernel, key=lambda k: -per_kernel[k].get("SQ_BUSY_CYCLES", per_kernel[k].get("SQ_INSTS_VALU", 0.0)))
    print("# sums over all dispatches of `bench.py --steps 1 --warmup 1 --no_cpu_baseline` (2 training steps), all XCDs")
    print("kernel".ljust(48) + "launches".rjust(9) + "".join(c.rjust(28) for c in counters))
    for k in keys[:24]:
        print(k[:47].ljust(48) + str(launches[k]).rjust(9) + "".join(("%.4g" % per_kernel[k].get(c, float("nan"))).rjust(28) for c in counters))
    print()
    print("# derived")
    for k in keys[:24]:
        v = per_kernel[k]
        out = []
        if v.get("SQ_INSTS_MFMA"):
            out.append("VALU(non-MFMA)/MFMA = %.2f" % ((v.get("SQ_INSTS_VALU", 0.0) - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"]))
            out.append("SALU/MFMA = %.2f" % (v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_MFMA"]))
        if v.get("SQ_BUSY_CYCLES") and v.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            out.append("MFMA busy / SQ busy = %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]))
        if out:
            print(k[:47].ljust(48) + "   ".join(out))


if __name__ == "__main__":
    main()
