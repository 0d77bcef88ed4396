"""Host-side helpers on the training path (reference utils/utilities.py): folders, logging, int16<->float,
the Mixup lambda generator.  The evaluation / sed_eval helpers of the reference are out of scope."""
import logging
import os
import pickle

import numpy as np


def create_folder(fd):
    if not os.path.exists(fd):
        os.makedirs(fd)


def get_filename(path):
    path = os.path.realpath(path)
    na_ext = path.split('/')[-1]
    return os.path.splitext(na_ext)[0]


def create_logging(log_dir, filemode):
    """utilities.py:29-51: <log_dir>/NNNN.log + console."""
    create_folder(log_dir)
    i1 = 0
    while os.path.isfile(os.path.join(log_dir, '{:04d}.log'.format(i1))):
        i1 += 1
    log_path = os.path.join(log_dir, '{:04d}.log'.format(i1))
    logging.basicConfig(level=logging.DEBUG,
                        format='%(asctime)s %(filename)s[line:%(lineno)d] %(levelname)s %(message)s',
                        datefmt='%a, %d %b %Y %H:%M:%S', filename=log_path, filemode=filemode)
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(logging.Formatter('%(name)-12s: %(levelname)-8s %(message)s'))
    logging.getLogger('').addHandler(console)
    return logging


def float32_to_int16(x):
    if np.max(np.abs(x)) > 1.:
        x = x / np.max(np.abs(x))
    return (x * 32767.).astype(np.int16)


def int16_to_float32(x):
    return (x / 32767.).astype(np.float32)


class Mixup(object):
    def __init__(self, mixup_alpha, random_seed=1234):
        """Mixup coefficient generator (utilities.py:220-242)."""
        self.mixup_alpha = mixup_alpha
        self.random_state = np.random.RandomState(random_seed)

    def get_lambda(self, batch_size):
        """-> (batch_size,) float64: [lam0, 1-lam0, lam1, 1-lam1, ...]."""
        lams = np.empty(batch_size, dtype=np.float64)
        for n in range(0, batch_size, 2):
            lam = self.random_state.beta(self.mixup_alpha, self.mixup_alpha, 1)[0]
            lams[n] = lam
            if n + 1 < batch_size:
                lams[n + 1] = 1. - lam
        return lams


class StatisticsContainer(object):
    """utilities.py:188-217 (pickle of per-iteration evaluation statistics)."""

    def __init__(self, statistics_path):
        self.statistics_path = statistics_path
        self.statistics_dict = {'test': [], 'evaluate': []}

    def append(self, data_type, iteration, statistics):
        statistics['iteration'] = iteration
        self.statistics_dict[data_type].append(statistics)

    def dump(self):
        pickle.dump(self.statistics_dict, open(self.statistics_path, 'wb'))
        logging.info('    Dump statistics to {}'.format(self.statistics_path))

    def load_state_dict(self, resume_iteration):
        self.statistics_dict = pickle.load(open(self.statistics_path, 'rb'))
        out = {'test': [], 'evaluate': []}
        for key in self.statistics_dict.keys():
            for statistics in self.statistics_dict[key]:
                if statistics['iteration'] <= resume_iteration:
                    out[key].append(statistics)
        self.statistics_dict = out
