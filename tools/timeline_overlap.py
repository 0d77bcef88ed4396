#!/usr/bin/env python
"""Digest a rocprofv3 --kernel-trace CSV: how much of the wall time do kernels of two HIP streams overlap, and which
kernels ran beside the weight-gradient kernels?   python tools/timeline_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    t0 = rows[0][0]
    wg = [(a, b, n) for a, b, n, q, s in rows if "wgrad_wino2_kernel" in n]
    other = [(a, b, n) for a, b, n, q, s in rows if "wgrad_wino2" not in n]
    busy = sum(b - a for a, b, *_ in rows)
    # union of intervals
    ivs = sorted((a, b) for a, b, *_ in rows)
    union, cur_a, cur_b = 0, ivs[0][0], ivs[0][1]
    for a, b in ivs[1:]:
        if a > cur_b:
            union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    print("kernels %d, sum of durations %.2f ms, union %.2f ms, overlapped %.2f ms, span %.2f ms"
          % (len(rows), busy / 1e6, union / 1e6, (busy - union) / 1e6, (rows[-1][1] - t0) / 1e6))
    beside = defaultdict(lambda: [0, 0.0, 0.0])
    j = 0
    for a, b, n in wg:
        for oa, ob, on in other:
            if ob <= a or oa >= b:
                continue
            ov = min(b, ob) - max(a, oa)
            e = beside[short(on)]
            e[0] += 1
            e[1] += ov / 1e6
            e[2] += (ob - oa) / 1e6
    for k, (c, ov, tot) in sorted(beside.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  beside wgrad: %-62s x%-4d overlap %.2f ms of its %.2f ms" % (k, c, ov, tot))
    qs = defaultdict(int)
    for a, b, n, q, s in rows:
        qs[(q, s)] += 1
    print("queues/streams:", dict(qs))


if __name__ == "__main__":
    main(sys.argv[1])
