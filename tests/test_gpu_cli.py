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
    assert set(ck.keys()) == {"iteration", "model", "optimizer"} and ck["iteration"] == 0
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
