#!/usr/bin/env python
"""Digest a rocprofv3 --kernel-trace CSV of bench.py into per-step numbers: kernel-busy time (union of kernel
intervals), idle gaps between kernels (launch-bound time) and the per-kernel table of the LAST `steps` steps, found by the
Adam kernel that ends every step.   python tools/step_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main(path, steps=5):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
    ends = [i for i, r in enumerate(rows) if "adam_amsgrad_kernel" in r[2]]
    if len(ends) < steps + 1:
        raise SystemExit("need >= %d optimiser steps in the trace, found %d" % (steps + 1, len(ends)))
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    span = sel[-1][1] - rows[lo - 1][1]
    ivs = sorted((a, b) for a, b, _ in sel)
    union, ca, cb = 0, ivs[0][0], ivs[0][1]
    gaps = []
    for a, b in ivs[1:]:
        if a > cb:
            union += cb - ca
            gaps.append(a - cb)
            ca, cb = a, b
        else:
            cb = max(cb, b)
    union += cb - ca
    print("steps %d: span %.3f ms/step, kernels busy (union) %.3f ms/step, idle between kernels %.3f ms/step (%d gaps/step, "
          "median %.1f us), kernels/step %.0f" % (steps, span / steps / 1e6, union / steps / 1e6, (span - union) / steps / 1e6,
                                                  len(gaps) // steps, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0,
                                                  len(sel) / steps))
    tab = defaultdict(lambda: [0, 0])
    for a, b, n in sel:
        tab[short(n)][0] += 1
        tab[short(n)][1] += b - a
    for k, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %-72s %5.1f launches/step %8.3f ms/step  %7.1f us avg" % (k, c / steps, t / steps / 1e6, t / c / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
