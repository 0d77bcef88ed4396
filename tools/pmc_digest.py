"""Digest rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic per launch.

Usage (GPU box; separate passes because FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "HBM"):
    cd /tmp; export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline
    python tools/pmc_digest.py gpurun_out/pmc_fetch gpurun_out/pmc_write [meta.json] > profiles/r01/pmc_traffic.json

meta.json (written ON THE GPU BOX by tools/collect_profiles.sh at collection time: kernel source hashes, workload) becomes the
`_meta` entry; bench.py refuses to quote a figure whose kernel sources have changed since (bench.pmc_traffic).

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts a wide (16 B/lane) coalesced read at half its
bytes (same guide), so 'fetch_bytes_x2' doubles it; every kernel here reads through 16-B loads or 16-B LDS-DMA.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    acc, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            acc[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return {k: (acc[k], len(n[k])) for k in acc}


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE") if len(sys.argv) > 2 else {}
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, nf = fetch[k]
        w, nw = write.get(k, (0.0, 1))
        out[k] = {"launches": nf, "fetch_bytes_raw_per_launch": round(f * 1024 / nf),
                  "fetch_bytes_x2_per_launch": round(2 * f * 1024 / nf),
                  "write_bytes_per_launch": round(w * 1024 / max(nw, 1))}
    if len(sys.argv) > 3:
        out["_meta"] = json.load(open(sys.argv[3]))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
