"""Throughput of the input pipeline (utils/data_generator.py: TrainSampler -> DCASE2017Task4Dataset -> collate_fn) on a
memory-mapped .npy pack, CPU only.  The GPU consumes 2 * 2650 = 5300 waveforms/s (3.4 GB/s of int16) per MI355X.
    python tools/loader_bench.py [--clips 256] [--batch 512] [--batches 8] [--workers 0 4]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd.utils.data_generator import (DCASE2017Task4Dataset, PinnedBatchLoader,  # noqa: E402
                                                                             TrainSampler, collate_fn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--workers", type=int, nargs="*", default=[0, 4])
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="sedpack_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rs = np.random.RandomState(0)
        np.save(os.path.join(root, "waveform.npy"), (rs.randn(args.clips, 320000) * 3276.7).astype(np.int16))
        np.save(os.path.join(root, "target.npy"), (rs.rand(args.clips, 17) < 0.2).astype(np.float32))
        np.save(os.path.join(root, "audio_name.npy"), np.array([("c%05d.wav" % i).encode() for i in range(args.clips)]))
        for keep in (True, False):
            for nw in args.workers:
                ds = DCASE2017Task4Dataset(keep_int16=keep)
                loader = torch.utils.data.DataLoader(ds, batch_sampler=TrainSampler(root, args.batch), collate_fn=collate_fn,
                                                     num_workers=nw, pin_memory=False)
                it = iter(loader)
                next(it)                                  # warm-up (worker start, page-in)
                t0 = time.time()
                nbytes = 0
                for _ in range(args.batches):
                    b = next(it)
                    nbytes += b["waveform"].nbytes
                dt = time.time() - t0
                print("%s waveforms, %d workers: %8.0f waveforms/s  %6.2f GB/s" %
                      ("int16" if keep else "fp32 ", nw, args.batches * args.batch / dt, nbytes / dt / 1e9))
                del it, loader
        for th in (1, 2, 4):
            it = iter(PinnedBatchLoader(root, TrainSampler(root, args.batch), threads=th))
            next(it)
            t0 = time.time()
            for _ in range(args.batches):
                next(it)
            dt = time.time() - t0
            print("PinnedBatchLoader int16, %d threads: %8.0f waveforms/s  %6.2f GB/s" %
                  (th, args.batches * args.batch / dt, args.batches * args.batch * 640000 / dt / 1e9))
            it.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
