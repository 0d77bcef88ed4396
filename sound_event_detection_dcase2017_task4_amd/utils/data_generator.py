"""Input pipeline feeding the hot path: same batch format and sampling order as reference utils/data_generator.py
(`DCASE2017Task4Dataset.__getitem__(meta)`, `TrainSampler`, `TestSampler`, `collate_fn`).

The packed store keeps the reference's dataset names (utils/features.py:232-260): `audio_name`, `waveform` int16
(N, 320000), `target` float32 (N, 17), optional `strong_target` bool (N, 1001, 17).  It can be an HDF5 file (needs
h5py, absent in the build image), a directory of `<name>.npy` files (memory-mapped), or an in-memory synthetic store.
"""
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


def open_store(path):
    """`synthetic:N[:samples]` | directory of .npy | .h5 file."""
    if isinstance(path, str) and path.startswith('synthetic:'):
        if path not in SyntheticStore._cache:
            parts = path.split(':')
            SyntheticStore._cache[path] = SyntheticStore(int(parts[1]), int(parts[2]) if len(parts) > 2 else 320000)
        return SyntheticStore._cache[path]
    if os.path.isdir(path):
        return _NpyStore(path)
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


def collate_fn(list_data_dict):
    """list of per-clip dicts -> dict of stacked numpy arrays."""
    return {key: np.array([d[key] for d in list_data_dict]) for key in list_data_dict[0].keys()}
