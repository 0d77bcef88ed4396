"""Post-processing / evaluator port against golden vectors produced by the reference's own
utils/vad.py and utilities.frame_prediction_to_event_prediction (tests/golden/make_golden_postproc.py), plus
known-answer tests of the segment-based metrics restated from sed_eval."""
import os

import numpy as np
import pytest
import torch

from sound_event_detection_dcase2017_task4_amd.utils import config, utilities as U, vad


@pytest.fixture(scope="module")
def post(golden_dir):
    return np.load(os.path.join(golden_dir, "postproc.npz"))


def test_activity_detection_matches_reference(post):
    off = 0
    for x, prm, c in zip(post["tracks"], post["params"], post["counts"]):
        ref = post["pairs"][off:off + c].tolist()
        off += c
        got = vad.activity_detection(x, prm[0], None if np.isnan(prm[1]) else prm[1], int(prm[2]), int(prm[3]))
        assert got == ref
    assert off == len(post["pairs"])


def test_activity_detection_edges():
    x = np.zeros(20, dtype=np.float32)
    assert vad.activity_detection(x, 0.5, 0.2, 10, 10) == []                       # nothing active
    x[:] = 0.9
    assert vad.activity_detection(x, 0.5, 0.2, 1, 0) == [[0, 20]]                  # everything active, low thres grows to the end
    assert vad.activity_detection(x, 0.5, None, 1, 0) == [[0, 19]]                 # last run ends ON its last frame (reference quirk)
    y = np.zeros(20, dtype=np.float32); y[3:6] = 0.9; y[19] = 0.9                  # a later run starting on the last frame
    assert vad.activity_detection(y, 0.5, 0.2, 0, 0)[0] == [3, 6]                  # (the reference raises IndexError here)


def test_frame_prediction_to_event_prediction_matches_reference(post):
    od = {"audio_name": post["ev_names"], "clipwise_output": post["ev_clip"],
          "framewise_output": np.repeat(post["ev_frame125"], 8, axis=1)}
    prm = {"audio_tagging_threshold": 0.5, "sed_high_threshold": 0.5, "sed_low_threshold": 0.2, "n_smooth": 10, "n_salt": 10}
    ev = U.frame_prediction_to_event_prediction(od, prm)
    assert prm["n_smooth"] == 10                                                    # caller's dict untouched
    names = list(post["ev_names"])
    got = np.array([[names.index(e["filename"]), e["onset"], e["offset"], config.lb_to_idx[e["event_label"]]] for e in ev])
    np.testing.assert_array_equal(got, post["ev_events"])
    per_class = dict(prm, sed_high_threshold=[0.5] * 17, n_salt=[10] * 17)          # per-class lists are accepted too
    assert U.frame_prediction_to_event_prediction(od, per_class) == ev


def test_submission_roundtrip_and_segment_metrics(tmp_path):
    ev = [{"filename": "Yabc.wav", "onset": 0.0, "offset": 1.0, "event_label": "Car"},
          {"filename": "Yabc.wav", "onset": 0.0, "offset": 2.0, "event_label": "Bus"}]
    sub = tmp_path / "sub.csv"
    U.write_submission(ev, str(sub))
    est = U.load_event_list(str(sub))
    assert [e["filename"] for e in est] == ["abc.wav", "abc.wav"] and est[1]["offset"] == 2.0      # leading 'Y' dropped
    ref_csv = tmp_path / "ref.csv"
    ref_csv.write_text("abc.wav\t0.0\t2.5\tCar\nabc.wav\t1.0\t2.0\tBus\nempty.wav\t\t\t\n")
    r = U.official_evaluate(str(ref_csv), str(sub))
    # hand computation (1 s segments): seg0 ref{Car} est{Car,Bus}: I=1; seg1 ref{Car,Bus} est{Bus}: D=1; seg2 ref{Car} est{}: D=1
    o = r["overall"]
    assert o["count"] == {"Nref": 4.0, "Nsys": 3.0}
    assert o["error_rate"]["error_rate"] == pytest.approx(0.75) and o["error_rate"]["substitution_rate"] == 0.0
    assert o["error_rate"]["deletion_rate"] == pytest.approx(0.5) and o["error_rate"]["insertion_rate"] == pytest.approx(0.25)
    assert o["f_measure"]["precision"] == pytest.approx(2 / 3) and o["f_measure"]["recall"] == pytest.approx(0.5)
    assert o["f_measure"]["f_measure"] == pytest.approx(4 / 7)
    assert r["class_wise"]["Car"]["f_measure"]["recall"] == pytest.approx(1 / 3)
    # a pure substitution: one reference label, one different estimated label in the same segment
    s = U.segment_based_metrics([{"filename": "f", "onset": 0.0, "offset": 1.0, "event_label": "A"}],
                                [{"filename": "f", "onset": 0.2, "offset": 0.9, "event_label": "B"}], event_label_list=["A", "B"])
    assert s["overall"]["error_rate"] == {"error_rate": 1.0, "substitution_rate": 1.0, "deletion_rate": 0.0, "insertion_rate": 0.0}
    # perfect system
    p = U.segment_based_metrics(U.load_event_list(str(ref_csv)), U.load_event_list(str(ref_csv)))
    assert p["overall"]["error_rate"]["error_rate"] == 0.0 and p["overall"]["f_measure"]["f_measure"] == 1.0


def test_evaluator_end_to_end_with_a_stub_model(tmp_path):
    """Evaluator.evaluate wiring (forward -> AP -> events -> submission -> segment metrics) with a stub model that
    returns the strong targets as its framewise output: perfect framewise AP, and events for every tagged class."""
    from sound_event_detection_dcase2017_task4_amd.pytorch.evaluate import Evaluator, sed_average_precision

    rs = np.random.RandomState(0)
    N = 6
    strong = np.zeros((N, 1000, 17), dtype=np.float32)
    for n in range(N):
        for k in rs.choice(17, 2, replace=False):
            b = rs.randint(0, 600)
            strong[n, b:b + rs.randint(100, 300), k] = 1.0
    strong[0, :400, 0] = 1.0
    for k in range(17):
        strong[1 + k % (N - 1), 500:650, k] = 1.0          # every class occurs (AP is undefined for absent classes)
    weak = strong.max(axis=1)
    names = np.array(["Yclip%d.wav" % n for n in range(N)])

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.i = 0

        def forward(self, wave):
            b = wave.shape[0]
            fr = torch.from_numpy(strong[self.i:self.i + b] * 0.9 + 0.05)
            self.i += b
            return {"clipwise_output": fr.max(dim=1)[0], "framewise_output": fr}

    loader = [{"audio_name": names[i:i + 3], "waveform": np.zeros((3, 8), dtype=np.float32), "target": weak[i:i + 3],
               "strong_target": strong[i:i + 3]} for i in (0, 3)]
    ref_csv = tmp_path / "ref.csv"
    with open(ref_csv, "w") as f:
        for n in range(N):
            for k in range(17):
                on = np.nonzero(np.diff(np.concatenate(([0], strong[n, :, k], [0]))))[0]
                for b, e in zip(on[::2], on[1::2]):
                    f.write("%s\t%f\t%f\t%s\n" % (names[n][1:], b / 100.0, e / 100.0, config.labels[k]))
    stats, out = Evaluator(Stub()).evaluate(loader, str(ref_csv), str(tmp_path / "sub.csv"))
    assert out["framewise_output"].shape == (N, 1000, 17)
    assert np.allclose(stats["clipwise_ap"], 1.0) and np.allclose(stats["framewise_ap"], 1.0)
    assert sed_average_precision(strong, out["framewise_output"], "macro") == pytest.approx(1.0)
    er = stats["sed_metrics"]["overall"]["error_rate"]["error_rate"]
    assert er < 0.1 and stats["sed_metrics"]["overall"]["f_measure"]["f_measure"] > 0.95


def test_segment_based_metrics_known_answers():
    """Eight hand-derived known-answer cases of the segment-based metrics (tests/golden/segment_metrics_cases.json: each
    carries its derivation): multi-label overlap, pure substitution, empty estimate, a file listed without reference
    events, an event straddling a segment boundary, an estimate outlasting the reference (roll padding), substitution +
    deletion inside one segment, accumulation over files with an unevaluated label.  Pinned to the PUBLISHED algorithm of
    sed_eval.sound_event.SegmentBasedMetrics (what utilities.py:142-185 of the reference calls) -- sed_eval itself is
    not installed, so not to its code."""
    import json
    from sound_event_detection_dcase2017_task4_amd.utils import utilities as U
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "segment_metrics_cases.json")))
    assert len(fx["cases"]) >= 8
    for c in fx["cases"]:
        r = U.segment_based_metrics(c["ref"], c["est"], time_resolution=1.0, event_label_list=c["labels"])
        o, w = r["overall"], c["want"]
        assert o["count"] == {"Nref": float(w["Nref"]), "Nsys": float(w["Nsys"])}, c["name"]
        nref = max(w["Nref"], 1)
        assert o["error_rate"]["substitution_rate"] == pytest.approx(w["S"] / nref), c["name"]
        assert o["error_rate"]["deletion_rate"] == pytest.approx(w["D"] / nref), c["name"]
        assert o["error_rate"]["insertion_rate"] == pytest.approx(w["I"] / nref), c["name"]
        assert o["error_rate"]["error_rate"] == pytest.approx(w["ER"]), c["name"]
        assert o["f_measure"]["precision"] == pytest.approx(w["P"]) and o["f_measure"]["recall"] == pytest.approx(w["R"]), c["name"]
        assert o["f_measure"]["f_measure"] == pytest.approx(w["F"]), c["name"]
        # class-wise counts add up to the overall ones; the class-wise average is the plain mean over the evaluated labels
        assert sum(v["count"]["Nref"] for v in r["class_wise"].values()) == w["Nref"]
        assert sum(v["count"]["Nsys"] for v in r["class_wise"].values()) == w["Nsys"]
        f_mean = np.mean([v["f_measure"]["f_measure"] for v in r["class_wise"].values()])
        assert r["class_wise_average"]["f_measure"]["f_measure"] == pytest.approx(f_mean)
        acc = o["accuracy"]
        assert 0.0 <= acc["accuracy"] <= 1.0 and acc["balanced_accuracy"] == pytest.approx(0.5 * (acc["sensitivity"] + acc["specificity"]))


def test_segment_based_metrics_against_an_independent_formulation():
    """sed_eval is not installed, so `utilities.segment_based_metrics` (what reference utilities.py:142-185 gets from
    sed_eval.sound_event.SegmentBasedMetrics) cannot be held to sed_eval's code.  Beside the hand-derived known answers it is held
    here to a SECOND, independently written evaluation of the published definitions (Mesaros, Heittola, Virtanen 2016) on 300 random
    event lists: segment activity by interval OVERLAP (a label is active in segment k iff an event has onset < k + 1 and offset > k)
    instead of floor / ceil slicing; micro and class-wise precision / recall / F from scikit-learn on the flattened segment x label
    matrices; substitutions / deletions / insertions per segment from the paper's FN / FP form (S = min(FN, FP), D = max(0, FN - FP),
    I = max(0, FP - FN)) instead of the Nref / Nsys / Ntp form.  Onsets and offsets fall on quarter seconds, so segment boundaries
    are hit exactly; files may be missing from the estimate, the estimate may name files and labels the reference does not have."""
    from sklearn.metrics import precision_recall_fscore_support
    from sound_event_detection_dcase2017_task4_amd.utils import utilities as U
    rs = np.random.RandomState(2017)
    labels_all = ["Car", "Bus", "Train horn", "Skateboard", "Bicycle"]
    for case in range(300):
        labels = labels_all[:rs.randint(2, 6)]
        files = ["f%d.wav" % i for i in range(rs.randint(1, 4))]

        def events(fnames, lab, p_file):
            out = []
            for f in fnames:
                if rs.rand() > p_file:
                    continue
                for _ in range(rs.randint(0, 5)):
                    on = rs.randint(0, 36) / 4.0
                    out.append({"filename": f, "onset": on, "offset": on + rs.randint(0, 13) / 4.0, "event_label": lab[rs.randint(len(lab))]})
            return out
        ref = events(files, labels, 1.0)
        for f in files:                                   # every reference file is LISTED (sed_eval evaluates the files of the reference)
            if not any(e["filename"] == f for e in ref):
                ref.append({"filename": f, "onset": 0.0, "offset": 0.0, "event_label": None})
        est = events(files + ["stranger.wav"], labels + ["Unknown"], 0.8)
        got = U.segment_based_metrics(ref, est, time_resolution=1.0, event_label_list=labels)
        # ---- the independent evaluation
        R_all, S_all, Ssub = [], [], [0.0, 0.0, 0.0]
        for f in files:
            r_ev = [e for e in ref if e["filename"] == f and e["event_label"] in labels]
            s_ev = [e for e in est if e["filename"] == f and e["event_label"] in labels]
            end = max([e["offset"] for e in r_ev + s_ev] + [0.0])
            nseg = int(np.ceil(end))
            for k in range(nseg):
                r = np.array([any(e["event_label"] == l and e["onset"] < k + 1 and e["offset"] > k for e in r_ev) for l in labels])
                s = np.array([any(e["event_label"] == l and e["onset"] < k + 1 and e["offset"] > k for e in s_ev) for l in labels])
                fn, fp = int((r & ~s).sum()), int((s & ~r).sum())
                Ssub[0] += min(fn, fp); Ssub[1] += max(0, fn - fp); Ssub[2] += max(0, fp - fn)
                R_all.append(r); S_all.append(s)
        R_all = np.array(R_all, dtype=int).reshape(-1, len(labels)); S_all = np.array(S_all, dtype=int).reshape(-1, len(labels))
        nref, nsys = float(R_all.sum()), float(S_all.sum())
        o = got["overall"]
        assert o["count"] == {"Nref": nref, "Nsys": nsys}, case
        d = nref if nref > 0 else 1.0
        assert o["error_rate"]["substitution_rate"] == pytest.approx(Ssub[0] / d) and o["error_rate"]["deletion_rate"] == pytest.approx(Ssub[1] / d)
        assert o["error_rate"]["insertion_rate"] == pytest.approx(Ssub[2] / d) and o["error_rate"]["error_rate"] == pytest.approx(sum(Ssub) / d)
        if R_all.size:
            p, r_, f1, _ = precision_recall_fscore_support(R_all.reshape(-1), S_all.reshape(-1), average="binary", zero_division=0)
            assert o["f_measure"]["precision"] == pytest.approx(p) and o["f_measure"]["recall"] == pytest.approx(r_), case
            assert o["f_measure"]["f_measure"] == pytest.approx(f1), case
            pc, rc, fc, _ = precision_recall_fscore_support(R_all, S_all, average=None, zero_division=0)
            for i, l in enumerate(labels):
                cw = got["class_wise"][l]
                assert cw["f_measure"]["f_measure"] == pytest.approx(fc[i]) and cw["f_measure"]["precision"] == pytest.approx(pc[i]), (case, l)
                assert cw["count"] == {"Nref": float(R_all[:, i].sum()), "Nsys": float(S_all[:, i].sum())}
                assert cw["error_rate"]["error_rate"] == pytest.approx(((R_all[:, i] & (1 - S_all[:, i])).sum() + (S_all[:, i] & (1 - R_all[:, i])).sum())
                                                                       / (R_all[:, i].sum() if R_all[:, i].sum() > 0 else 1.0))
