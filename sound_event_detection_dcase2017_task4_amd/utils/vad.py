"""Frame probabilities -> [onset, offset) frame pairs (reference utils/vad.py:12-134, called from
utils/utilities.py:109-114).

Host-side numpy, same results as the reference INCLUDING its index conventions, which the submission files and hence
the published metrics depend on:
  * runs of frames above `thres` (vad.py:43-64): the first run starts at its first frame, every LATER run is reported
    from its second frame (`loc + 1`), every run but the last ends one past its last frame, the LAST run ends ON its
    last frame;
  * the low threshold grows a run in both directions while the probability stays >= low_thres (:67-89), then runs
    closer than 1 frame are merged;
  * `smooth` merges runs separated by <= n_smooth frames (:92-116); `remove_salt_noise` drops runs of length
    <= n_salt (:119-134).
Pinned by tests/golden/postproc.npz (vectors produced by importing the reference's vad.py).
"""
import numpy as np


def find_bgn_fin_pairs(locts):
    locts = np.asarray(locts)
    if locts.size == 0:
        return []
    brk = np.nonzero(np.diff(locts) > 1)[0]                  # run k ends at locts[brk[k]], run k+1 starts after it
    bgns = np.concatenate(([locts[0]], locts[brk + 1] + 1))
    fins = np.concatenate((locts[brk] + 1, [locts[-1]]))
    return [[int(b), int(f)] for b, f in zip(bgns, fins)]


def smooth(bgn_fin_pairs, n_smooth):
    if len(bgn_fin_pairs) == 0:
        return []
    out = []
    cur_bgn = bgn_fin_pairs[0][0]
    for (_, prev_fin), (bgn, _) in zip(bgn_fin_pairs[:-1], bgn_fin_pairs[1:]):
        if bgn - prev_fin > n_smooth:
            out.append([cur_bgn, prev_fin])
            cur_bgn = bgn
    out.append([cur_bgn, bgn_fin_pairs[-1][1]])
    return out


def remove_salt_noise(bgn_fin_pairs, n_salt):
    return [[b, f] for b, f in bgn_fin_pairs if f - b > n_salt]


def activity_detection_with_second_thres(x, bgn_fin_pairs, thres):
    n = len(x)
    grown = []
    for bgn, fin in bgn_fin_pairs:
        bgn = min(bgn, n - 1)      # a later run reported from `loc + 1` may point one past the end (the reference
        while bgn != -1 and not x[bgn] < thres:           # raises IndexError there; clamp instead)
            bgn -= 1
        while fin != n and not x[fin] < thres:
            fin += 1
        grown.append([bgn + 1, fin])
    return smooth(grown, n_smooth=1)


def activity_detection(x, thres, low_thres=None, n_smooth=1, n_salt=0):
    x = np.asarray(x)
    pairs = find_bgn_fin_pairs(np.nonzero(x > thres)[0])
    if low_thres is not None:
        pairs = activity_detection_with_second_thres(x, pairs, low_thres)
    pairs = smooth(pairs, n_smooth)
    return remove_salt_noise(pairs, n_salt)
