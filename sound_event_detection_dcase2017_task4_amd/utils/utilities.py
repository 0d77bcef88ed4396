"""Host-side helpers on the training path (reference utils/utilities.py): folders, logging, int16<->float,
the Mixup lambda generator, and the post-processing / scoring helpers of the evaluation branch (events from framewise
probabilities, submission files, segment-based metrics restated from sed_eval, which is not installed)."""
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
    precision / recall / F accumulate over all segments of all files that appear in the reference.  PARITY UNPINNED
    against sed_eval itself (known-answer tests only).  Returns the same nested dict layout as sed_eval's results()."""
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
    tot = dict(Ntp=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0, S=0.0, D=0.0, I=0.0)
    cw = {l: dict(Ntp=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0) for l in event_label_list}

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
        tp, fp, fn = (r + s > 1), (s - r > 0), (r - s > 0)
        nref, nsys, ntp = r.sum(1), s.sum(1), tp.sum(1)
        tot['Ntp'] += ntp.sum(); tot['Nfp'] += fp.sum(); tot['Nfn'] += fn.sum(); tot['Nref'] += nref.sum(); tot['Nsys'] += nsys.sum()
        tot['S'] += (np.minimum(nref, nsys) - ntp).sum()
        tot['D'] += np.maximum(0, nref - nsys).sum()
        tot['I'] += np.maximum(0, nsys - nref).sum()
        for l, i in lab.items():
            c = cw[l]
            c['Ntp'] += tp[:, i].sum(); c['Nfp'] += fp[:, i].sum(); c['Nfn'] += fn[:, i].sum()
            c['Nref'] += r[:, i].sum(); c['Nsys'] += s[:, i].sum()

    def prf(ntp, nref, nsys):
        p = ntp / nsys if nsys > 0 else 0.0
        r_ = ntp / nref if nref > 0 else 0.0
        return {'f_measure': 2 * p * r_ / (p + r_) if p + r_ > 0 else 0.0, 'precision': p, 'recall': r_}

    def er(S, D, I, nref):
        d = nref if nref > 0 else 1.0
        return {'error_rate': (S + D + I) / d, 'substitution_rate': S / d, 'deletion_rate': D / d, 'insertion_rate': I / d}

    class_wise = {}
    for l, c in cw.items():
        d_, i_ = c['Nfn'], c['Nfp']
        class_wise[l] = {'f_measure': prf(c['Ntp'], c['Nref'], c['Nsys']),
                         'error_rate': {'error_rate': (d_ + i_) / (c['Nref'] if c['Nref'] > 0 else 1.0),
                                        'deletion_rate': d_ / (c['Nref'] if c['Nref'] > 0 else 1.0),
                                        'insertion_rate': i_ / (c['Nref'] if c['Nref'] > 0 else 1.0)},
                         'count': {'Nref': c['Nref'], 'Nsys': c['Nsys']}}
    return {'overall': {'f_measure': prf(tot['Ntp'], tot['Nref'], tot['Nsys']),
                        'error_rate': er(tot['S'], tot['D'], tot['I'], tot['Nref']),
                        'count': {'Nref': tot['Nref'], 'Nsys': tot['Nsys']}},
            'class_wise': class_wise}


def official_evaluate(reference_csv_path, prediction_csv_path):
    """utilities.py:142-185 with sed_eval replaced by segment_based_metrics (1 s resolution)."""
    return segment_based_metrics(load_event_list(reference_csv_path), load_event_list(prediction_csv_path), time_resolution=1.0)
