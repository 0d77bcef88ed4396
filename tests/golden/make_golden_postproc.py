"""Golden vectors for the post-processing port: run the REFERENCE's utils/vad.py (pure numpy, importable here) on
seeded probability tracks.  Usage (this container only): python tests/golden/make_golden_postproc.py
Writes tests/golden/postproc.npz: tracks (N, 1000), params (N, 4) = thres, low_thres (nan = None), n_smooth, n_salt,
and the reference's [bgn, fin] pairs flattened with per-track counts."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/utils")
import vad  # noqa: E402  (the reference module)

rs = np.random.RandomState(2017)
tracks, params, flat, counts = [], [], [], []
for n in range(160):
    T = 1000
    kind = n % 4
    if kind == 0:                                  # smooth bumps
        t = np.arange(T)
        x = np.zeros(T)
        for _ in range(rs.randint(1, 6)):
            c, w, a = rs.randint(0, T), rs.randint(3, 120), rs.rand()
            x = np.maximum(x, a * np.exp(-0.5 * ((t - c) / w) ** 2))
    elif kind == 1:                                # piecewise constant (frame x8 interpolation look)
        x = np.repeat(rs.rand(125), 8)
    elif kind == 2:                                # noisy
        x = np.clip(np.repeat(rs.rand(125), 8) + rs.randn(T) * 0.1, 0, 1)
    else:                                          # sparse spikes, edges active
        x = (rs.rand(T) < 0.05).astype(np.float64) * rs.rand(T)
        x[:rs.randint(0, 5)] = 0.9
        x[T - rs.randint(0, 5):] = 0.9
    thres = [0.5, 0.3, 0.7][n % 3]
    low = [0.2, np.nan, 0.1, 0.45][(n // 3) % 4]
    n_smooth = [10, 1, 0, 25][(n // 5) % 4]
    n_salt = [10, 0, 3, 40][(n // 7) % 4]
    try:
        pairs = vad.activity_detection(x.astype(np.float32), thres, None if np.isnan(low) else low, n_smooth, n_salt)
    except IndexError:          # the reference indexes x[len(x)] when a later run starts on the very last frame
        continue
    tracks.append(x.astype(np.float32)); params.append([thres, low, n_smooth, n_salt])
    counts.append(len(pairs)); flat += [list(p) for p in pairs]
# ---- frame_prediction_to_event_prediction (utilities.py:70-121).  utilities.py itself cannot be imported here (it
# imports librosa / h5py / sed_eval at module level), so the one function is compiled from the reference file's AST
# into a namespace holding the reference's own config and vad.activity_detection.
import ast  # noqa: E402
import config as ref_config  # noqa: E402
src = open("/root/reference/utils/utilities.py").read()
fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "frame_prediction_to_event_prediction"][0]
ns = {"config": ref_config, "activity_detection": vad.activity_detection, "np": np}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "utilities.py", "exec"), ns)
N = 24
names = np.array(["Y%06d_%d.000_%d.000.wav" % (i * 37, i, i + 10) for i in range(N)])
clip = rs.rand(N, 17).astype(np.float32)
frame = np.clip(np.repeat(rs.rand(N, 125, 17), 8, axis=1) * np.repeat(clip[:, None, :], 1000, axis=1) * 1.6, 0, 1).astype(np.float32)
od = {"audio_name": names, "clipwise_output": clip, "framewise_output": frame}
sed_params = {"audio_tagging_threshold": 0.5, "sed_high_threshold": 0.5, "sed_low_threshold": 0.2, "n_smooth": 10, "n_salt": 10}
events = ns["frame_prediction_to_event_prediction"](od, dict(sed_params))
ev = np.array([[list(names).index(e["filename"]), e["onset"], e["offset"], ref_config.lb_to_idx[e["event_label"]]] for e in events],
              dtype=np.float64)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "postproc.npz")
np.savez_compressed(out, tracks=np.stack(tracks), params=np.array(params, dtype=np.float64),
                    pairs=np.array(flat, dtype=np.int64).reshape(-1, 2), counts=np.array(counts, dtype=np.int64),
                    ev_names=names, ev_clip=clip, ev_frame125=frame[:, ::8, :], ev_events=ev)
print(out, sum(counts), "pairs", len(events), "events")
