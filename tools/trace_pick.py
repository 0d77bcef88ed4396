#!/usr/bin/env python
"""Per-dispatch durations of the kernels whose name contains one of the given substrings, last optimiser step of a rocprofv3
--kernel-trace CSV:   python tools/trace_pick.py <kernel_trace.csv> gemm gru ..."""
import csv
import sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")),
               r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))) for r in csv.DictReader(open(sys.argv[1])))
ends = [i for i, r in enumerate(rows) if "adam_amsgrad_kernel" in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = rows[lo][0]
for a, b, n, g, w in rows[lo:hi]:
    if any(k in n for k in sys.argv[2:]):
        print("%9.1f us  +%8.1f us  grid %8s wg %4s  %s" % ((a - t0) / 1e3, (b - a) / 1e3, g, w, n.replace("(anonymous namespace)::", "")[:90]))
