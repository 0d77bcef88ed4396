"""Host-side helpers on the training path (reference utils/utilities.py): folders, logging, int16<->float,
the Mixup lambda generator, and the post-processing / scoring helpers of the evaluation branch (events from framewise
probabilities, submission files, segment-based metrics restated from sed_eval, which is not installed)."""
import datetime
import logging
import os
import pickle
import re

import numpy as np


def create_folder(fd):
    os.makedirs(fd, exist_ok=True)


def get_filename(path):
    """'/a/b/main.py' -> 'main' (symlinks resolved; names the workspace sub-folders, main.py:72)."""
    return os.path.splitext(os.path.basename(os.path.realpath(path)))[0]


def create_logging(log_dir, filemode):
    """Log DEBUG+ to the next free <log_dir>/NNNN.log and INFO+ to the console (what utilities.py:29-51 sets up).
    Handlers are attached to the root logger explicitly, so a second call (tests, several runs in one process) opens
    a new file instead of being ignored the way a repeated logging.basicConfig() is."""
    create_folder(log_dir)
    taken = [int(m.group(1)) for m in (re.match(r"^(\d{4})\.log$", f) for f in os.listdir(log_dir)) if m]
    index = 0
    while index in taken:                          # first gap, like the reference's isfile() probe
        index += 1
    root = logging.getLogger('')
    for h in [h for h in root.handlers if getattr(h, '_sed_handler', False)]:
        root.removeHandler(h)
        h.close()
    root.setLevel(logging.DEBUG)
    to_file = logging.FileHandler(os.path.join(log_dir, '{:04d}.log'.format(index)), mode=filemode)
    to_file.setLevel(logging.DEBUG)
    to_file.setFormatter(logging.Formatter('%(asctime)s %(filename)s[line:%(lineno)d] %(levelname)s %(message)s',
                                           datefmt='%a, %d %b %Y %H:%M:%S'))
    to_console = logging.StreamHandler()
    to_console.setLevel(logging.INFO)
    to_console.setFormatter(logging.Formatter('%(name)-12s: %(levelname)-8s %(message)s'))
    for h in (to_file, to_console):
        h._sed_handler = True
        root.addHandler(h)
    return logging


def float32_to_int16(x):
    """[-1, 1] float waveform -> int16 storage (peak-normalised first when it overshoots; utilities.py:61-64)."""
    x = np.asarray(x)
    peak = float(np.abs(x).max()) if x.size else 0.0
    return ((x / peak if peak > 1.0 else x) * 32767.0).astype(np.int16)


def int16_to_float32(x):
    """utilities.py:66-67.  (On the GPU path the division is folded into the log-mel kernel's load instead.)"""
    return (np.asarray(x) / 32767.0).astype(np.float32)


class Mixup(object):
    """Mixup coefficient stream (utilities.py:220-242): one Beta(alpha, alpha) draw per PAIR of batch rows from a
    RandomState(seed), emitted as [lam0, 1-lam0, lam1, 1-lam1, ...].  The legacy RandomState fills an array draw by
    draw, so one `beta(size=pairs)` call consumes the stream exactly like the reference's per-pair scalar draws
    (tests/test_capi_and_host.py pins this bit for bit)."""

    def __init__(self, mixup_alpha, random_seed=1234):
        self.mixup_alpha = mixup_alpha
        self.random_state = np.random.RandomState(random_seed)

    def get_lambda(self, batch_size):
        pairs = (batch_size + 1) // 2
        lam = self.random_state.beta(self.mixup_alpha, self.mixup_alpha, size=pairs)
        return np.stack([lam, 1.0 - lam], axis=1).reshape(-1)[:2 * pairs].astype(np.float64)


def random_state_to_plain(rs):
    """numpy RandomState -> a dict of a torch tensor and Python scalars (what a checkpoint may hold and still load under
    torch.load(weights_only=True))."""
    import torch
    name, keys, pos, has_gauss, cached = rs.get_state()
    return {'name': str(name), 'keys': torch.from_numpy(np.asarray(keys, dtype=np.int64).copy()), 'pos': int(pos),
            'has_gauss': int(has_gauss), 'cached_gaussian': float(cached)}


def random_state_from_plain(rs, d):
    rs.set_state((d['name'], np.asarray(d['keys'], dtype=np.uint32), int(d['pos']), int(d['has_gauss']), float(d['cached_gaussian'])))
    return rs


class StatisticsContainer(object):
    """Evaluation statistics of a run, appended every 1000 iterations and pickled (utilities.py:188-217): a dict
    {'train': [...], 'test': [...], 'evaluate': [...]} of per-iteration dicts, written to `statistics_path` and to a
    time-stamped backup next to it.  load_state_dict(resume_iteration) keeps the entries up to that iteration."""
    DATA_TYPES = ('train', 'test', 'evaluate')

    def __init__(self, statistics_path):
        self.statistics_path = statistics_path
        stem = os.path.splitext(statistics_path)[0]
        self.backup_statistics_path = '{}_{}.pkl'.format(stem, datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S'))
        self.statistics_dict = {k: [] for k in self.DATA_TYPES}

    def append(self, data_type, iteration, statistics):
        statistics['iteration'] = iteration
        self.statistics_dict[data_type].append(statistics)

    def dump(self):
        for path in (self.statistics_path, self.backup_statistics_path):
            with open(path, 'wb') as f:
                pickle.dump(self.statistics_dict, f)
            logging.info('    Dump statistics to {}'.format(path))

    def load_state_dict(self, resume_iteration):
        with open(self.statistics_path, 'rb') as f:
            stored = pickle.load(f)
        self.statistics_dict = {k: [st for st in stored.get(k, []) if st['iteration'] <= resume_iteration]
                                for k in set(self.DATA_TYPES) | set(stored)}


# ---------------------------------------------------------------------------------------------------------------
# Post-processing and segment-based scoring (reference utils/utilities.py:70-185).  Host-side numpy.

def frame_prediction_to_event_prediction(output_dict, sed_params_dict):
    """Framewise probabilities -> list of {'filename', 'onset', 'offset', 'event_label'} (utilities.py:70-121).

    A class is searched for events in a clip only if its clipwise probability exceeds the audio-tagging threshold;
    the framewise track then goes through vad.activity_detection (high / low threshold, smoothing, salt removal).
    Every entry of `sed_params_dict` may be a float (all classes) or a per-class list; the dict is not modified."""
    from . import config
    from .vad import activity_detection
    audios_num, frames_num, classes_num = output_dict['framewise_output'].shape

    def per_class(v):
        return list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * classes_num

    at_thres = per_class(sed_params_dict['audio_tagging_threshold'])
    hi, lo = per_class(sed_params_dict['sed_high_threshold']), per_class(sed_params_dict['sed_low_threshold'])
    n_smooth, n_salt = per_class(sed_params_dict['n_smooth']), per_class(sed_params_dict['n_salt'])
    fps = float(config.frames_per_second)
    event_list = []
    for n in range(audios_num):
        for k in np.nonzero(output_dict['clipwise_output'][n] > np.asarray(at_thres))[0]:
            for bgn, fin in activity_detection(output_dict['framewise_output'][n, :, k], thres=hi[k], low_thres=lo[k],
                                               n_smooth=n_smooth[k], n_salt=n_salt[k]):
                event_list.append({'filename': output_dict['audio_name'][n], 'onset': bgn / fps, 'offset': fin / fps,
                                   'event_label': config.labels[k]})
    return event_list


def write_submission(event_list, submission_path):
    """Tab-separated submission file (utilities.py:124-139).  The first character of the audio name (the 'Y' the
    packing step prepends) is dropped, as in the reference, so that names match the ground-truth csv."""
    with open(submission_path, 'w') as f:
        for event in event_list:
            f.write('{}\t{}\t{}\t{}\n'.format(str(event['filename'])[1:], event['onset'], event['offset'], event['event_label']))
    logging.info('    Write submission file to {}'.format(submission_path))


def load_event_list(csv_path):
    """Rows 'filename<TAB>onset<TAB>offset<TAB>event_label' -> list of dicts; rows without times (files with no event)
    only register the file name."""
    events = []
    with open(csv_path) as f:
        for line in f:
            parts = line.rstrip('\n').split('\t')
            if len(parts) >= 4 and parts[1] != '' and parts[2] != '':
                events.append({'filename': parts[0], 'onset': float(parts[1]), 'offset': float(parts[2]),
                               'event_label': parts[3]})
            elif parts[0]:
                events.append({'filename': parts[0], 'onset': None, 'offset': None, 'event_label': None})
    return events


def segment_based_metrics(reference_event_list, estimated_event_list, time_resolution=1.0, event_label_list=None):
    """Segment-based error rate and F-score (what official_evaluate, utilities.py:142-185, obtains from the third-party
    sed_eval.sound_event.SegmentBasedMetrics with time_resolution=1.0).

    sed_eval is not installed here or on the GPU box, so this is a restatement of its published algorithm (Mesaros,
    Heittola, Virtanen: "Metrics for polyphonic sound event detection", 2016; sed_eval 0.2.x): per file both event
    lists become binary segment x label rolls (onset floored, offset ceiled to the resolution, the shorter roll
    zero-padded); per segment Ntp / Nfp / Nfn give substitutions S = min(Nref, Nsys) - Ntp, deletions
    D = max(0, Nref - Nsys), insertions I = max(0, Nsys - Nref); ER = (S + D + I) / Nref and micro-averaged
    precision / recall / F accumulate over all segments of all files that appear in the reference.  Pinned to the
    PUBLISHED algorithm by hand-derived known-answer vectors (tests/golden/segment_metrics_cases.json); unpinned against
    sed_eval's own code.  Returns the nested dict layout of sed_eval's results(): 'overall' (f_measure, error_rate,
    accuracy, count), 'class_wise', 'class_wise_average'."""
    def by_file(events):
        d = {}
        for e in events:
            d.setdefault(e['filename'], [])
            if e.get('event_label') is not None:
                d[e['filename']].append(e)
        return d

    ref_files, est_files = by_file(reference_event_list), by_file(estimated_event_list)
    if event_label_list is None:
        event_label_list = sorted({e['event_label'] for e in reference_event_list if e.get('event_label') is not None})
    lab = {l: i for i, l in enumerate(event_label_list)}
    L = len(event_label_list)
    tot = dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0, S=0.0, D=0.0, I=0.0)
    cw = {l: dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0) for l in event_label_list}

    def roll(events):
        events = [e for e in events if e['event_label'] in lab]
        if not events:
            return np.zeros((0, L))
        n = int(np.ceil(max(e['offset'] for e in events) / time_resolution))
        r = np.zeros((n, L))
        for e in events:
            r[int(np.floor(e['onset'] / time_resolution)):int(np.ceil(e['offset'] / time_resolution)), lab[e['event_label']]] = 1
        return r

    for fname in sorted(ref_files):
        r, s = roll(ref_files[fname]), roll(est_files.get(fname, []))
        n = max(len(r), len(s))
        r = np.vstack([r, np.zeros((n - len(r), L))]); s = np.vstack([s, np.zeros((n - len(s), L))])
        tp, tn, fp, fn = (r + s > 1), (r + s == 0), (s - r > 0), (r - s > 0)
        nref, nsys, ntp = r.sum(1), s.sum(1), tp.sum(1)
        tot['Ntp'] += ntp.sum(); tot['Ntn'] += tn.sum(); tot['Nfp'] += fp.sum(); tot['Nfn'] += fn.sum()
        tot['Nref'] += nref.sum(); tot['Nsys'] += nsys.sum()
        tot['S'] += (np.minimum(nref, nsys) - ntp).sum()
        tot['D'] += np.maximum(0, nref - nsys).sum()
        tot['I'] += np.maximum(0, nsys - nref).sum()
        for l, i in lab.items():
            c = cw[l]
            c['Ntp'] += tp[:, i].sum(); c['Ntn'] += tn[:, i].sum(); c['Nfp'] += fp[:, i].sum(); c['Nfn'] += fn[:, i].sum()
            c['Nref'] += r[:, i].sum(); c['Nsys'] += s[:, i].sum()

    def prf(ntp, nref, nsys):
        p = ntp / nsys if nsys > 0 else 0.0
        r_ = ntp / nref if nref > 0 else 0.0
        return {'f_measure': 2 * p * r_ / (p + r_) if p + r_ > 0 else 0.0, 'precision': p, 'recall': r_}

    def er(S, D, I, nref):
        d = nref if nref > 0 else 1.0
        return {'error_rate': (S + D + I) / d, 'substitution_rate': S / d, 'deletion_rate': D / d, 'insertion_rate': I / d}

    def acc(c):
        sens = c['Ntp'] / (c['Ntp'] + c['Nfn']) if c['Ntp'] + c['Nfn'] > 0 else 0.0
        spec = c['Ntn'] / (c['Ntn'] + c['Nfp']) if c['Ntn'] + c['Nfp'] > 0 else 0.0
        n = c['Ntp'] + c['Ntn'] + c['Nfp'] + c['Nfn']
        return {'accuracy': (c['Ntp'] + c['Ntn']) / n if n > 0 else 0.0, 'balanced_accuracy': 0.5 * (sens + spec),
                'sensitivity': sens, 'specificity': spec}

    class_wise = {}
    for l, c in cw.items():
        d_, i_ = c['Nfn'], c['Nfp']
        class_wise[l] = {'f_measure': prf(c['Ntp'], c['Nref'], c['Nsys']),
                         'error_rate': {'error_rate': (d_ + i_) / (c['Nref'] if c['Nref'] > 0 else 1.0),
                                        'deletion_rate': d_ / (c['Nref'] if c['Nref'] > 0 else 1.0),
                                        'insertion_rate': i_ / (c['Nref'] if c['Nref'] > 0 else 1.0)},
                         'accuracy': acc(c),
                         'count': {'Nref': c['Nref'], 'Nsys': c['Nsys']}}

    def mean_of(group, key):
        vals = [v[group][key] for v in class_wise.values()]
        return float(np.mean(vals)) if vals else 0.0

    average = {'f_measure': {k: mean_of('f_measure', k) for k in ('f_measure', 'precision', 'recall')},
               'error_rate': {k: mean_of('error_rate', k) for k in ('error_rate', 'deletion_rate', 'insertion_rate')},
               'accuracy': {k: mean_of('accuracy', k) for k in ('accuracy', 'balanced_accuracy', 'sensitivity', 'specificity')}}
    return {'overall': {'f_measure': prf(tot['Ntp'], tot['Nref'], tot['Nsys']),
                        'error_rate': er(tot['S'], tot['D'], tot['I'], tot['Nref']),
                        'accuracy': acc(tot),
                        'count': {'Nref': tot['Nref'], 'Nsys': tot['Nsys']}},
            'class_wise': class_wise, 'class_wise_average': average}


def official_evaluate(reference_csv_path, prediction_csv_path):
    """utilities.py:142-185 with sed_eval replaced by segment_based_metrics (1 s resolution)."""
    return segment_based_metrics(load_event_list(reference_csv_path), load_event_list(prediction_csv_path), time_resolution=1.0)
