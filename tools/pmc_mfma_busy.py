"""Digest a `rocprofv3 --kernel-trace --pmc MfmaUtil` pass into per-kernel-family MFMA pipe-busy shares (min / max over the
template variants of a family, each averaged over its dispatches).
    python tools/pmc_mfma_busy.py gpurun_out/prof_r05/pmc_b32_sqD "bs=32 training step" [meta.json] > profiles/r05/pmc_mfma_busy_b32.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = ("conv_sf16_kernel", "wgrad_sf16_kernel", "conv_wino2_kernel", "wgrad_wino2_kernel", "gemm_sf16_kernel", "gemm_tn_sf16_kernel")


def main():
    d, what = sys.argv[1], sys.argv[2]
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "MfmaUtil":
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    out = {}
    for fam in FAMILIES:
        per_variant = {k: sum(v) / len(v) / 100.0 for k, v in acc.items() if k.startswith(fam)}
        if per_variant:
            out[fam] = {"mfma_busy_frac_min": round(min(per_variant.values()), 4), "mfma_busy_frac_max": round(max(per_variant.values()), 4),
                        "variants": {k: [round(v, 4), len(acc[k])] for k, v in sorted(per_variant.items())},
                        "definition": "MfmaUtil (rocprofv3 --pmc, derived counter: MFMA-busy cycles / active cycles) averaged over the "
                                      "dispatches of each kernel variant inside the %s; min / max over the variants; "
                                      "variants = {name: [share, dispatches]}" % what}
    if len(sys.argv) > 3:
        out["_meta"] = json.load(open(sys.argv[3]))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
