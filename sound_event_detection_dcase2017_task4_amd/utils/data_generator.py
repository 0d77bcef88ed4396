"""Input pipeline feeding the hot path: same batch format and sampling order as reference utils/data_generator.py
(`DCASE2017Task4Dataset.__getitem__(meta)`, `TrainSampler`, `TestSampler`, `collate_fn`).

The packed store keeps the reference's dataset names (utils/features.py:232-260): `audio_name`, `waveform` int16
(N, 320000), `target` float32 (N, 17), optional `strong_target` bool (N, 1001, 17).  It can be an HDF5 file (needs
h5py, absent in the build image), a directory of `<name>.npy` files (memory-mapped), or an in-memory synthetic store.
"""
import collections
import logging
import os

import numpy as np

from .utilities import int16_to_float32


class _NpyStore(object):
    def __init__(self, path):
        self.path = path
        self.arrays = {}
        for f in os.listdir(path):
            if f.endswith('.npy'):
                self.arrays[f[:-4]] = np.load(os.path.join(path, f), mmap_mode='r')

    def keys(self):
        return self.arrays.keys()

    def __getitem__(self, k):
        return self.arrays[k]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class SyntheticStore(object):
    """N(0, 0.1^2) int16 clips + Bernoulli(0.2) weak targets, generated once, held in memory."""
    _cache = {}

    def __init__(self, n_clips, audio_samples=320000, seed=1234):
        rs = np.random.RandomState(seed)
        w = np.clip(rs.randn(n_clips, audio_samples) * 0.1, -1, 1)
        self.arrays = {'audio_name': np.array([('syn_%05d.wav' % i).encode() for i in range(n_clips)]),
                       'waveform': (w * 32767.).astype(np.int16),
                       'target': (rs.rand(n_clips, 17) < 0.2).astype(np.float32)}

    def keys(self):
        return self.arrays.keys()

    def __getitem__(self, k):
        return self.arrays[k]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NPY_CACHE = {}


def open_store(path):
    """`synthetic:N[:samples]` | directory of .npy | .h5 file."""
    if isinstance(path, str) and path.startswith('synthetic:'):
        if path not in SyntheticStore._cache:
            parts = path.split(':')
            SyntheticStore._cache[path] = SyntheticStore(int(parts[1]), int(parts[2]) if len(parts) > 2 else 320000)
        return SyntheticStore._cache[path]
    if os.path.isdir(path):
        # memory maps are opened once per process (the reference re-opens its HDF5 file for every clip,
        # data_generator.py:37; at >5000 waveforms/s per GPU that alone would need several workers)
        key = os.path.abspath(path)
        if key not in _NPY_CACHE:
            _NPY_CACHE[key] = _NpyStore(path)
        return _NPY_CACHE[key]
    try:
        import h5py
    except ImportError:
        raise RuntimeError("reading %s needs h5py, which is not installed; convert the pack to a directory of .npy "
                           "files (same dataset names) or use 'synthetic:N'" % path)
    return h5py.File(path, 'r')


class DCASE2017Task4Dataset(object):
    def __init__(self, keep_int16=False):
        """keep_int16=True hands the raw int16 waveform to the GPU (the log-mel kernel folds the /32767)."""
        self.keep_int16 = keep_int16

    def __getitem__(self, meta):
        """meta: {'hdf5_path': ..., 'index_in_hdf5': i} -> {'audio_name', 'waveform', 'target', ('strong_target')}"""
        index = meta['index_in_hdf5']
        with open_store(meta['hdf5_path']) as hf:
            name = hf['audio_name'][index]
            audio_name = name.decode() if isinstance(name, bytes) else str(name)
            wav = np.asarray(hf['waveform'][index])
            waveform = wav if self.keep_int16 else int16_to_float32(wav)
            data_dict = {'audio_name': audio_name, 'waveform': waveform,
                         'target': np.asarray(hf['target'][index]).astype(np.float32)}
            if 'strong_target' in hf.keys():
                data_dict['strong_target'] = np.asarray(hf['strong_target'][index]).astype(np.float32)
        return data_dict


class TrainSampler(object):
    def __init__(self, hdf5_path, batch_size, random_seed=1234):
        """Infinite shuffled stream of batch metas; seed 1234; reshuffle at wrap-around (data_generator.py:52-101,
        including its double indexing `audio_indexes[audio_indexes[pointer]]`)."""
        self.hdf5_path = hdf5_path
        self.batch_size = batch_size
        self.random_state = np.random.RandomState(random_seed)
        with open_store(hdf5_path) as hf:
            self.audios_num = len(hf['audio_name'])
        logging.info('Training audio num: {}'.format(self.audios_num))
        self.audio_indexes = np.arange(self.audios_num)
        self.random_state.shuffle(self.audio_indexes)
        self.pointer = 0

    def skip(self, n_batches):
        """Advance the stream by n_batches without building them (resume: the stream continues where the checkpointed run was;
        same pointer walk and reshuffles as __iter__)."""
        todo = int(n_batches) * self.batch_size
        while todo > 0:
            step = min(todo, self.audios_num - self.pointer)
            self.pointer += step
            todo -= step
            if self.pointer >= self.audios_num:
                self.pointer = 0
                self.random_state.shuffle(self.audio_indexes)

    def __iter__(self):
        while True:
            batch_meta = []
            for _ in range(self.batch_size):
                index = self.audio_indexes[self.pointer]
                self.pointer += 1
                if self.pointer >= self.audios_num:
                    self.pointer = 0
                    self.random_state.shuffle(self.audio_indexes)
                batch_meta.append({'hdf5_path': self.hdf5_path, 'index_in_hdf5': self.audio_indexes[index]})
            yield batch_meta


class TestSampler(object):
    def __init__(self, hdf5_path, batch_size):
        self.hdf5_path = hdf5_path
        self.batch_size = batch_size
        with open_store(hdf5_path) as hf:
            self.audios_num = len(hf['audio_name'])
        logging.info('Test audio num: {}'.format(self.audios_num))
        self.audio_indexes = np.arange(self.audios_num)

    def __iter__(self):
        pointer = 0
        while pointer < self.audios_num:
            idx = np.arange(pointer, min(pointer + self.batch_size, self.audios_num))
            yield [{'hdf5_path': self.hdf5_path, 'index_in_hdf5': self.audio_indexes[i]} for i in idx]
            pointer += self.batch_size


class ShardedBatchSampler(object):
    """Rows [row_lo, row_hi) of every batch of `sampler`: the slice of ONE global sampler stream that a data-parallel rank
    consumes (the reference scatters each batch over its GPUs, main.py:138; here every rank walks the same seeded stream and
    loads only its own rows)."""

    def __init__(self, sampler, row_lo, row_hi):
        self.sampler, self.row_lo, self.row_hi = sampler, int(row_lo), int(row_hi)
        self.batch_size = self.row_hi - self.row_lo

    def __iter__(self):
        for batch_meta in self.sampler:
            yield batch_meta[self.row_lo:self.row_hi]


def collate_fn(list_data_dict):
    """list of per-clip dicts -> dict of stacked numpy arrays."""
    return {key: np.array([d[key] for d in list_data_dict]) for key in list_data_dict[0].keys()}


class PinnedBatchLoader(object):
    """Drop-in for `DataLoader(DCASE2017Task4Dataset(keep_int16=True), batch_sampler=..., collate_fn=collate_fn)` that
    keeps up with an MI355X (5300 waveforms/s = 3.4 GB/s of int16 per GPU at 2650 clips/s with mixup).

    The reference pipeline (data_generator.py:15-164 + torch DataLoader workers) re-opens the pack per clip, stacks the
    batch with np.array and pickles the 328 MB result through a worker pipe: measured 310 waveforms/s with 4 workers.
    Here the batches of the SAME sampler stream are assembled by a few THREADS (numpy row copies release the GIL) straight
    from the memory-mapped pack into a ring of page-locked int16 buffers; with `device` set the upload runs on a copy
    stream one batch ahead and the batch dict holds device tensors.

    Yields {'audio_name': list[str], 'waveform': int16 (B2, L) tensor, 'target': float32 (B2, 17) tensor,
    ['strong_target']}; the tensors of a batch are valid until `hold` + 1 further batches have been requested (hold = 0: until
    the next one; the train CLI holds 3 so that it can re-run the batches of refused optimiser steps)."""

    def __init__(self, hdf5_path, batch_sampler, device=None, depth=3, threads=4, hold=0):
        import torch
        self.torch = torch
        self.store = open_store(hdf5_path)
        self.sampler = batch_sampler
        self.device = device
        self.hold = max(0, int(hold))
        self.depth = max(2, int(depth), self.hold + 3)
        self.threads = max(1, int(threads))
        self.wave_src = self.store['waveform']
        self.target_src = self.store['target']
        self.names = self.store['audio_name']
        self.strong_src = self.store['strong_target'] if 'strong_target' in self.store.keys() else None
        self._slots = None

    def _alloc(self, n):
        torch = self.torch
        L = self.wave_src.shape[1]
        pin = self.device is not None or torch.cuda.is_available()
        slots = []
        for _ in range(self.depth):
            s = {'wave': torch.empty((n, L), dtype=torch.int16, pin_memory=pin),
                 'target': torch.empty((n, self.target_src.shape[1]), dtype=torch.float32, pin_memory=pin)}
            if self.strong_src is not None:
                s['strong'] = torch.empty((n,) + tuple(self.strong_src.shape[1:]), dtype=torch.float32, pin_memory=pin)
            if self.device is not None:
                s['dwave'] = torch.empty((n, L), dtype=torch.int16, device=self.device)
                s['dtarget'] = torch.empty((n, self.target_src.shape[1]), dtype=torch.float32, device=self.device)
                s['ready'] = torch.cuda.Event()
                s['consumed'] = torch.cuda.Event()
                s['in_use'] = False
            slots.append(s)
        return slots

    def __iter__(self):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        torch = self.torch
        q = queue.Queue()
        free_slots = threading.Semaphore(self.depth)   # a slot is busy from fill start until the consumer asks for the batch after it
        stop = threading.Event()
        copy_stream = torch.cuda.Stream(device=self.device) if self.device is not None else None
        pool = ThreadPoolExecutor(max_workers=self.threads)

        def fill(slot, idx):
            # (rows beyond len(idx) of a slot keep stale data; the consumer only sees the first len(idx) rows)
            wave_np, target_np = slot['wave'].numpy(), slot['target'].numpy()
            strong_np = slot['strong'].numpy() if 'strong' in slot else None

            def rows(lo, hi):
                for r in range(lo, hi):
                    wave_np[r] = self.wave_src[idx[r]]
                    target_np[r] = self.target_src[idx[r]]
                    if strong_np is not None:
                        strong_np[r] = self.strong_src[idx[r]]
            n = len(idx)
            step = (n + self.threads - 1) // self.threads
            list(pool.map(lambda lo: rows(lo, min(n, lo + step)), range(0, n, step)))

        def producer():
            try:
                if self.device is not None:
                    # page-locked allocations and the copy stream belong to THIS rank's GPU (a thread starts with device
                    # 0 current, which would create a context on GPU 0 from every rank)
                    torch.cuda.set_device(self.device)
                i = 0
                for batch_meta in self.sampler:
                    if stop.is_set():
                        return
                    idx = [int(m['index_in_hdf5']) for m in batch_meta]
                    # slots are sized once, for the sampler's full batch; a shorter (last) batch uses their first rows
                    cap = max(len(idx), int(getattr(self.sampler, 'batch_size', 0) or 0))
                    if self._slots is None or self._slots[0]['wave'].shape[0] < len(idx):
                        self._slots = self._alloc(cap)
                    while not free_slots.acquire(timeout=0.2):
                        if stop.is_set():
                            return
                    slot = self._slots[i % self.depth]
                    if self.device is not None and slot['in_use']:
                        slot['ready'].synchronize()          # the previous upload out of this pinned buffer is done
                    fill(slot, idx)
                    names = [self.names[j] for j in idx]
                    names = [x.decode() if isinstance(x, bytes) else str(x) for x in names]
                    n = len(idx)
                    if self.device is not None:
                        with torch.cuda.stream(copy_stream):
                            if slot['in_use']:
                                copy_stream.wait_event(slot['consumed'])   # its last readers have been enqueued and ran
                            slot['dwave'][:n].copy_(slot['wave'][:n], non_blocking=True)
                            slot['dtarget'][:n].copy_(slot['target'][:n], non_blocking=True)
                            slot['ready'].record(copy_stream)
                        slot['in_use'] = True
                    q.put((slot, names, n))
                    i += 1
                q.put(None)
            except BaseException as e:                        # surface loader errors in the consumer
                q.put(e)

        t = threading.Thread(target=producer, daemon=True)
        t.start()
        prev = None
        held = collections.deque()                    # slots handed out and still promised to the consumer (oldest first)
        try:
            first = True
            while True:
                if not first:
                    # the batch handed out `hold` + 1 requests ago is no longer needed: mark the point in the consumer's stream
                    # after its last reader FIRST, then let the producer have the slot (its upload waits for that event)
                    held.append(prev)
                    while len(held) > self.hold:
                        old = held.popleft()
                        if old is not None:
                            old['consumed'].record(torch.cuda.current_stream(self.device))
                        free_slots.release()
                first = False
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                slot, names, n = item
                if self.device is not None:
                    torch.cuda.current_stream(self.device).wait_event(slot['ready'])
                    out = {'audio_name': names, 'waveform': slot['dwave'][:n], 'target': slot['dtarget'][:n]}
                    prev = slot
                else:
                    out = {'audio_name': names, 'waveform': slot['wave'][:n], 'target': slot['target'][:n]}
                if 'strong' in slot:
                    out['strong_target'] = slot['strong'][:n]
                yield out
        finally:
            stop.set()
            try:
                while True:
                    q.get_nowait()
            except Exception:
                pass
            t.join(timeout=5.0)
            pool.shutdown(wait=True)
