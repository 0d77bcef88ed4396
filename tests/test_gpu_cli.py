"""GPU: the train / inference_prob CLI end to end on synthetic clips (checkpoint format, pickled outputs)."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_then_inference_cli(tmp_path):
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli
    ws = str(tmp_path)
    common = ["--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", "Cnn_9layers_FrameAtt",
              "--loss_type", "clip_bce", "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "12"]
    cli.main(["train"] + common + ["--learning_rate", "1e-3", "--resume_iteration", "0", "--stop_iteration", "2",
                                   "--print_every", "1"])
    ck_dir = os.path.join(ws, "checkpoints", "main", "holdout_fold=1", "model_type=Cnn_9layers_FrameAtt",
                          "loss_type=clip_bce", "augmentation=mixup", "batch_size=4")
    ck = torch.load(os.path.join(ck_dir, "0_iterations.pth"), map_location="cpu")
    # the reference's three keys (main.py:222-230) + 'streams' (additive: what a resumed run continues its random streams from)
    assert set(ck.keys()) == {"iteration", "model", "optimizer", "streams"} and ck["iteration"] == 0
    assert "att_block.bn_att.weight" in ck["model"] and ck["model"]["conv_block1.conv1.weight"].shape == (64, 1, 3, 3)
    cli.main(["inference_prob"] + common + ["--iteration", "0"])
    pred = pickle.load(open(os.path.join(ws, "predictions", "main", "holdout_fold=1", "model_type=Cnn_9layers_FrameAtt",
                                         "loss_type=clip_bce", "augmentation=mixup", "batch_size=4",
                                         "0_iterations.prediction.test.pkl"), "rb"))
    assert pred["clipwise_output"].shape == (12, 17) and pred["framewise_output"].shape == (12, 1000, 17)
    assert np.isfinite(pred["clipwise_output"]).all() and pred["target"].shape == (12, 17)


def test_pinned_batch_loader_device_mode_delivers_intact_batches(tmp_path):
    """PinnedBatchLoader with a device: uploads run one batch ahead on a copy stream into recycled buffers; every batch
    must still equal what the reference pipeline (DataLoader + collate_fn) yields for the same sampler stream, also when
    the consumer's stream is busy and lags behind the producer."""
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import (DCASE2017Task4Dataset, PinnedBatchLoader,
                                                                                 TrainSampler, collate_fn)
    rs = np.random.RandomState(1)
    N, L = 23, 32000
    np.save(tmp_path / "waveform.npy", (rs.randn(N, L) * 3000).astype(np.int16))
    np.save(tmp_path / "target.npy", (rs.rand(N, 17) < 0.2).astype(np.float32))
    np.save(tmp_path / "audio_name.npy", np.array([("c%03d.wav" % i).encode() for i in range(N)]))
    root = str(tmp_path)
    dev = torch.device("cuda", 0)
    ref = iter(torch.utils.data.DataLoader(DCASE2017Task4Dataset(keep_int16=True), batch_sampler=TrainSampler(root, 8),
                                           collate_fn=collate_fn, num_workers=0))
    busy = torch.randn(2048, 2048, device=dev)
    sums = []
    expect = []
    it = iter(PinnedBatchLoader(root, TrainSampler(root, 8), device=dev, depth=2, threads=2))
    for k in range(10):
        b = next(it)
        a = next(ref)
        assert b["waveform"].device.type == "cuda" and b["waveform"].dtype == torch.int16
        for _ in range(3):
            busy = busy @ busy * 1e-3                    # keep the consumer stream behind the producer
        sums.append((b["waveform"].double().sum() + busy[0, 0] * 0, b["target"].sum()))   # reads enqueued after the matmuls
        expect.append((float(a["waveform"].astype(np.float64).sum()), float(a["target"].sum())))
        assert [str(x) for x in a["audio_name"]] == b["audio_name"]
    it.close()
    torch.cuda.synchronize()
    for (sw, st), (ew, et) in zip(sums, expect):
        assert float(sw) == ew and float(st) == et


def test_evaluator_end_to_end_through_the_hip_model(tmp_path):
    """reference pytorch/evaluate.py:52-89 driven by the HIP model on a small pack WITH strong labels: eval forward over a
    PinnedBatchLoader (partial last batch), clipwise / framewise average precision, events from the framewise track,
    submission file, segment-based metrics -- every number compared with the same pipeline fed by the CPU oracle's outputs."""
    from sklearn import metrics
    from oracle import model as om
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    from sound_event_detection_dcase2017_task4_amd.pytorch.evaluate import Evaluator, sed_average_precision
    from sound_event_detection_dcase2017_task4_amd.utils import config, utilities as U
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import PinnedBatchLoader, TestSampler
    mt = "Cnn_9layers_FrameAtt"
    rs = np.random.RandomState(8)
    N, L = 5, 320000
    wave = (np.clip(rs.randn(N, L) * 0.1, -1, 1) * 32767).astype(np.int16)
    strong = np.zeros((N, 1001, 17), dtype=bool)
    rows = []
    for n in range(N):
        for k in rs.choice(17, 2, replace=False):
            on = float(rs.randint(0, 6)); off = on + float(rs.randint(1, 4))
            strong[n, int(round(on * 100)):int(round(off * 100)) + 1, k] = True
            rows.append("c%02d.wav\t%.3f\t%.3f\t%s" % (n, on, off, config.labels[k]))
    pack = tmp_path / "testing"
    pack.mkdir()
    np.save(pack / "waveform.npy", wave)
    np.save(pack / "target.npy", strong.any(axis=1).astype(np.float32))
    np.save(pack / "strong_target.npy", strong)
    np.save(pack / "audio_name.npy", np.array([("Yc%02d.wav" % n).encode() for n in range(N)]))   # the packer prepends 'Y'
    csv = tmp_path / "groundtruth.csv"
    csv.write_text("\n".join(rows) + "\n")
    st = om.recipe_state(mt, 4)
    # a head that fires: otherwise no class passes the 0.5 tagging threshold and the event branch is never exercised
    st["att_block.cla.bias"] = st["att_block.cla.bias"] + 0.8
    m = getattr(models, mt)(32000, 1024, 320, 64, 50, 14000, 17)
    m.load_state_dict(st)
    m = m.to("cuda")
    loader = PinnedBatchLoader(str(pack), TestSampler(str(pack), batch_size=2), device=torch.device("cuda", 0))
    stats, out = Evaluator(model=m).evaluate(loader, str(csv), str(tmp_path / "sub.csv"))
    assert out["clipwise_output"].shape == (N, 17) and out["framewise_output"].shape == (N, 1000, 17)
    assert out["strong_target"].shape == (N, 1001, 17) and list(out["audio_name"]) == ["Yc%02d.wav" % n for n in range(N)]
    with torch.no_grad():
        ref = om.forward(mt, st, torch.from_numpy((wave / 32767.0).astype(np.float32)), training=False)
    rc, rf = ref["clipwise_output"].numpy(), ref["framewise_output"].numpy()
    assert np.abs(out["clipwise_output"] - rc).max() < 1e-4 and np.abs(out["framewise_output"] - rf).max() < 1e-4
    tgt = strong.any(axis=1).astype(np.float32)
    seen = tgt.sum(0) > 0                                           # AP is undefined for classes without positives
    # the scores are functions of the model outputs (equal to the oracle's within 1e-4, checked above): rank-based metrics on
    # 5 clips flip with the 7th digit of a probability, so they are recomputed here from the outputs the Evaluator saw
    hc, hf = out["clipwise_output"], out["framewise_output"]
    np.testing.assert_allclose(stats["clipwise_ap"][seen], metrics.average_precision_score(tgt, hc, average=None)[seen], atol=1e-9)
    np.testing.assert_allclose(stats["framewise_ap"][seen], sed_average_precision(strong[:, :1000].astype(np.float32), hf, None)[seen],
                               atol=1e-9)
    np.testing.assert_allclose(stats["framewise_ap"][seen], sed_average_precision(strong[:, :1000].astype(np.float32), rf, None)[seen],
                               atol=5e-2)
    ev_ref = U.frame_prediction_to_event_prediction({"audio_name": out["audio_name"], "clipwise_output": hc, "framewise_output": hf},
                                                    Evaluator(model=m).sed_params_dict)
    assert len(ev_ref) > 0
    want = U.segment_based_metrics(U.load_event_list(str(csv)),
                                   [dict(e, filename=e["filename"][1:]) for e in ev_ref], time_resolution=1.0)
    got = stats["sed_metrics"]
    assert abs(got["overall"]["error_rate"]["error_rate"] - want["overall"]["error_rate"]["error_rate"]) < 1e-9
    assert abs(got["overall"]["f_measure"]["f_measure"] - want["overall"]["f_measure"]["f_measure"]) < 1e-9
    assert got["overall"]["count"]["Nref"] == want["overall"]["count"]["Nref"] > 0


def test_train_cli_survives_a_non_finite_batch(tmp_path, monkeypatch, caplog):
    """A batch whose log-mel holds a NaN (diverged input): the Adam kernel refuses the poisoned update(s) on the device, the
    train loop gets ops.NonFiniteOperand at a later poll, logs it, switches to the fp32 MFMA kernels and re-runs the batch it
    is at -- the run finishes with finite parameters (the reference would have trained on with NaN weights)."""
    import logging
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli
    real = ops.logmel
    calls = {"n": 0}

    def poisoned(wave, tables, amin=1e-10):
        out = real(wave, tables, amin)
        calls["n"] += 1
        if calls["n"] == 2:
            out[0, 3, 5] = float("nan")
        return out

    monkeypatch.setattr(ops, "logmel", poisoned)
    monkeypatch.setattr(ops, "USE_SF16", True)
    ws = str(tmp_path)
    args = ["train", "--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", "Cnn_9layers_FrameAvg",
            "--loss_type", "clip_bce", "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "12",
            "--learning_rate", "1e-3", "--resume_iteration", "0", "--stop_iteration", "5", "--print_every", "1"]
    with caplog.at_level(logging.WARNING):
        cli.main(args)
    assert calls["n"] >= 6
    assert ops.USE_SF16 is False                                       # the loop switched to the fp32 kernels ...
    assert any("optimiser step(s) were refused" in r.getMessage() for r in caplog.records)      # ... and said so
    ops.check_device_errors(synchronize=True)                          # nothing left pending


def test_train_cli_with_hip_graph_prints_the_same_losses(tmp_path, capsys):
    """`train --hip_graph` (and the default `auto` at <= 8 clips per GPU): three eager iterations, then forward + loss + backward
    replayed as one HIP graph per iteration; the loss series is the eager run's (`--hip_graph off`)."""
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli
    series = []
    for extra in (["--hip_graph", "off"], ["--hip_graph"], []):
        ws = str(tmp_path / ("run%d" % len(series)))
        os.makedirs(ws)
        torch.manual_seed(4321)
        cli.main(["train", "--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", "Cnn_9layers_FrameAvg",
                  "--loss_type", "clip_bce", "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "12",
                  "--learning_rate", "1e-3", "--resume_iteration", "0", "--stop_iteration", "7", "--print_every", "1"] + extra)
        out = capsys.readouterr().out
        series.append([float(l.split()[1]) for l in out.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit()])
    assert len(series[0]) == len(series[1]) == len(series[2]) == 8
    np.testing.assert_allclose(series[1], series[0], rtol=5e-6, atol=0)
    assert series[2] == series[1]                      # auto = on at 4 clips per GPU: the very same replayed kernels


def test_train_cli_auto_graph_falls_back_to_the_eager_loop_when_capture_fails(tmp_path, capsys, monkeypatch, caplog):
    """`--hip_graph auto` is a default, not a request: a refused capture (a runtime this code has not seen) must cost the graph, not
    the run -- the CLI logs it and continues kernel by kernel with the same losses; `--hip_graph on` fails loudly."""
    import logging
    from sound_event_detection_dcase2017_task4_amd import graph
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli

    def args(ws, *extra):
        return ["train", "--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", "Cnn_9layers_FrameAvg",
                "--loss_type", "clip_bce", "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "12",
                "--learning_rate", "1e-3", "--resume_iteration", "0", "--stop_iteration", "5", "--print_every", "1"] + list(extra)

    def losses():
        out = capsys.readouterr().out
        return [float(l.split()[1]) for l in out.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit()]

    ws = str(tmp_path / "off"); os.makedirs(ws)
    torch.manual_seed(99)
    cli.main(args(ws, "--hip_graph", "off"))
    want = losses()

    real_body = graph.GraphedTrainStep._body

    def refuse(self):                                 # fails INSIDE the capture region only (the eager warm-up steps run)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("capture refused (test)")
        return real_body(self)
    monkeypatch.setattr(graph.GraphedTrainStep, "_body", refuse)
    ws = str(tmp_path / "auto"); os.makedirs(ws)
    torch.manual_seed(99)
    with caplog.at_level(logging.WARNING):
        cli.main(args(ws))
    got = losses()
    assert len(got) == len(want) == 6
    np.testing.assert_allclose(got, want, rtol=5e-6, atol=0)
    assert any("HIP graph capture failed" in r.getMessage() for r in caplog.records)
    ws = str(tmp_path / "on"); os.makedirs(ws)
    with pytest.raises(graph.GraphCaptureError, match="capture refused"):
        cli.main(args(ws, "--hip_graph", "on"))


def test_train_cli_auto_graph_does_not_swallow_errors_of_applied_steps(tmp_path, monkeypatch, caplog):
    """Only a refused CAPTURE may cost the graph (graph.GraphCaptureError: nothing of the step has run).  An exception out of a
    step that has been applied -- here optimizer.step() behind the first graph replay -- must propagate: re-running that batch
    eagerly would update Adam and the BatchNorm statistics twice and log the real error as a capture warning (round-5 advisor)."""
    import logging
    from sound_event_detection_dcase2017_task4_amd import optim
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli
    real_step = optim.FusedAdamAmsgrad.step
    calls = {"n": 0}

    def step(self):
        calls["n"] += 1
        if calls["n"] == 4:                           # three eager warm-up steps, then the first replay's optimiser step
            raise RuntimeError("boom behind the replay (test)")
        return real_step(self)
    monkeypatch.setattr(optim.FusedAdamAmsgrad, "step", step)
    ws = str(tmp_path)
    with caplog.at_level(logging.WARNING):
        with pytest.raises(RuntimeError, match="boom behind the replay"):
            cli.main(["train", "--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", "Cnn_9layers_FrameAvg",
                      "--loss_type", "clip_bce", "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "12",
                      "--learning_rate", "1e-3", "--resume_iteration", "0", "--stop_iteration", "5", "--print_every", "1"])
    assert calls["n"] == 4                            # the batch was NOT run again
    assert not any("HIP graph capture failed" in r.getMessage() for r in caplog.records)
    torch.cuda.synchronize()


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_Gru_FrameAtt"])
def test_train_cli_resume_continues_the_run_bit_for_bit(tmp_path, capsys, monkeypatch, mt):
    """reference pytorch/main.py:125-134 (resume; broken there: undefined name, optimiser and streams restart) and :221-231
    (checkpoint).  Here six iterations straight == three iterations + checkpoint + `--resume_iteration 3` + three more, bit for bit:
    parameters, Adam first / second moments, the amsgrad maximum, the step counter, every BatchNorm buffer, and the three streams
    (seed-1234 sampler, seed-1234 mixup lambdas, SpecAugment draws of the global torch generator) -- the losses printed for
    iterations 3..5 are the straight run's.  The checkpoint still loads under torch.load's default weights_only=True."""
    from sound_event_detection_dcase2017_task4_amd import optim
    from sound_event_detection_dcase2017_task4_amd.pytorch import main as cli
    kept = []

    class Recording(optim.FusedAdamAmsgrad):
        def __init__(self, model, *a, **k):
            super().__init__(model, *a, **k)
            kept.append((self, model))
    monkeypatch.setattr(cli, "FusedAdamAmsgrad", Recording)
    ws = str(tmp_path)
    common = ["train", "--dataset_dir", ws, "--workspace", ws, "--holdout_fold", "1", "--model_type", mt, "--loss_type", "clip_bce",
              "--augmentation", "mixup", "--batch_size", "4", "--cuda", "--synthetic", "14", "--learning_rate", "1e-3",
              "--print_every", "1", "--hip_graph", "off", "--checkpoint_every", "3"]

    def losses():
        out = capsys.readouterr().out
        return {int(l.split()[0]): float(l.split()[1]) for l in out.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit()}

    def snapshot():
        opt, model = kept[-1]
        torch.cuda.synchronize()
        st = {"flat": opt.flat.clone(), "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(), "vmax": opt.max_exp_avg_sq.clone(),
              "step": opt.step_count}
        st.update({"buf:" + k: v.clone() for k, v in model.named_buffers()})
        return st

    torch.manual_seed(2024)
    cli.main(common + ["--resume_iteration", "0", "--stop_iteration", "5"])          # iterations 0..5; checkpoints at 0 and 3
    straight, l_straight = snapshot(), losses()
    assert sorted(l_straight) == [0, 1, 2, 3, 4, 5]
    ck_dir = os.path.join(ws, "checkpoints", "main", "holdout_fold=1", "model_type=" + mt, "loss_type=clip_bce", "augmentation=mixup",
                          "batch_size=4")
    ck = torch.load(os.path.join(ck_dir, "3_iterations.pth"), map_location="cpu")        # default weights_only=True
    assert set(ck.keys()) == {"iteration", "model", "optimizer", "streams"} and ck["iteration"] == 3 and ck["optimizer"]["step"] == 3
    torch.manual_seed(1)                                   # a different generator state: the stream must come from the checkpoint
    cli.main(common + ["--resume_iteration", "3", "--stop_iteration", "5"])          # iterations 3..5
    resumed, l_resumed = snapshot(), losses()
    assert sorted(l_resumed) == [3, 4, 5]
    assert [l_resumed[i] for i in (3, 4, 5)] == [l_straight[i] for i in (3, 4, 5)]
    assert resumed["step"] == straight["step"] == 6
    for k in straight:
        if k != "step":
            assert torch.equal(straight[k], resumed[k]), k
    assert len(kept) == 2 and kept[0][0] is not kept[1][0]
