"""CPU: the C-ABI library loads and exports every symbol include/sed_hip.h declares (no compute calls without a
GPU), plus the host-side logic of the product (SpecAugment draws, Mixup lambdas, state_dict layout, utilities)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import model as om


def test_header_symbols_all_exported():
    from sound_event_detection_dcase2017_task4_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 38 and "sed_conv3x3_igemm" in protos and "sed_logmel_f32" in protos
    h = _lib.lib()
    for name in protos:
        assert hasattr(h, name), name
    # and nothing is exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (sed_\w+)", out))
    hooks = set(_lib.parse_header(_lib.TEST_HEADER))        # test hooks live in their own header, outside the product ABI
    assert hooks == {"sed_gru_set_spin_limit", "sed_gru_force_agent_scope", "sed_debug_occupy",
                     "sed_test_wgrad_sf16_reduce"} and not hooks & set(protos)
    assert exported == set(protos) | hooks, exported ^ (set(protos) | hooks)
    # ... and no product module calls a hook (tests / tools reach them through _lib.test_hooks())
    root = os.path.dirname(_lib.__file__)
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py") and f != "_lib.py":
                src = open(os.path.join(dp, f)).read()
                assert not any(k in src for k in hooks) and "test_hooks" not in src, os.path.join(dp, f)


def test_host_only_entry_points():
    from sound_event_detection_dcase2017_task4_amd import _lib
    h = _lib.lib()
    assert h.sed_version().startswith(b"sed-hip")
    assert h.sed_stats_rows_per_part() == 1024 and h.sed_conv1_rows_per_part() == 256
    assert h.sed_conv_rows_per_part(1000, 64) == 32 and h.sed_conv_rows_per_part(1000, 512) == 64
    assert h.sed_conv_rows_per_part(1 << 20, 64) == 32 and h.sed_conv_num_parts(1 << 20, 64) == 4 * 8192
    assert h.sed_conv_num_parts(1000, 128) == 16 and h.sed_conv_num_parts(1000, 64) == 32
    ns, pps = ctypes.c_int(), ctypes.c_int()
    M = 256 * 1001 * 64
    n = h.sed_wgrad_partial_floats(M, 64, 64, 9, ctypes.byref(ns), ctypes.byref(pps))
    assert ns.value * pps.value >= M and pps.value % 32 == 0 and pps.value <= 16384
    assert n == ns.value * 9 * 64 * 64


def test_argument_errors_are_reported_not_crashes():
    """Bad sizes are rejected on the host before any launch (error behaviour of the boundary)."""
    from sound_event_detection_dcase2017_task4_amd import _lib
    h = _lib.lib()
    assert h.sed_conv3x3_igemm(None, None, None, 1, 8, 8, 48, 64, None, None, 0, None, None, None, None, None, None, None) == -22
    assert h.sed_logmel_f32(None, 1, 100, None, None, None, 105, None, 4, None, 866, 1e-10, None, None) == -22     # L <= 512
    assert h.sed_gru_seq_fwd(None, None, None, None, None, 4, 5, 128, None, None, None, None, None, None) == -22        # hidden size != 256
    assert h.sed_mixup_rows(None, None, 3, 17, None, None) == -22
    assert h.sed_adam_amsgrad(None, None, None, None, None, 10, 0, 1e-3, 0.9, 0.999, 1e-8, 1.0, None, None, None, None, None, None) == -22
    assert h.sed_act_amax(None, 4, 64, None, None, None, None) == -22


def test_product_refuses_cpu_tensors():
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    m = models.Cnn_9layers_FrameAvg(32000, 1024, 320, 64, 50, 14000, 17)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 32000))
    with pytest.raises(RuntimeError):
        do_mixup(torch.zeros(4, 17), torch.ones(4))
    with pytest.raises(Exception):
        models.Cnn_9layers_FrameAvg(32000, 2048, 320, 64, 50, 14000, 17)      # kernels are specialised: loud failure


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sound_event_detection_dcase2017_task4_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(dp, f)


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_state_dict_layout_is_reference_compatible(mt):
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    m = getattr(models, mt)(32000, 1024, 320, 64, 50, 14000, 17)
    sd = m.state_dict()
    lay = om.state_layout(mt)
    assert list(sd.keys()) == [k for k, _ in lay]
    for k, shape in lay:
        assert tuple(sd[k].shape) == tuple(shape), k
    m.load_state_dict(om.recipe_state(mt, 1))                       # a reference-shaped checkpoint loads
    frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == sorted(om.FROZEN_KEYS)
    np.testing.assert_allclose(sd["logmel_extractor.melW"].numpy(), ofe.mel_matrix(), atol=2e-9)
    wr, wi = ofe.dft_weights()
    np.testing.assert_allclose(sd["spectrogram_extractor.stft.conv_real.weight"].numpy(), wr, atol=1e-6)
    np.testing.assert_allclose(sd["spectrogram_extractor.stft.conv_imag.weight"].numpy(), wi, atol=1e-6)


def test_specaug_fast_draw_equals_package_order():
    from sound_event_detection_dcase2017_task4_amd.utils.augmentation import draw_specaug_stripes
    for seed, (B, T) in enumerate([(1, 101), (6, 101), (64, 1001), (512, 1001)]):
        torch.manual_seed(seed)
        want = ofe.draw_specaug_stripes(B, T, 64)
        after_want = torch.randint(0, 1000, (4,))
        torch.manual_seed(seed)
        got = draw_specaug_stripes(B, T, 64)
        after_got = torch.randint(0, 1000, (4,))
        np.testing.assert_array_equal(got, want)
        assert torch.equal(after_want, after_got)                  # the generator is left in the same state


def test_mixup_generator_and_utilities(golden_dir, tmp_path):
    from sound_event_detection_dcase2017_task4_amd.utils.utilities import (Mixup, float32_to_int16, int16_to_float32,
                                                                          create_folder, get_filename)
    misc = np.load(os.path.join(golden_dir, "misc.npz"))
    np.testing.assert_array_equal(Mixup(1.).get_lambda(64), misc["mixup_lambda64"])
    g = Mixup(1.)
    a, b = g.get_lambda(4), g.get_lambda(4)                        # the stream continues across calls (main.py:233-235)
    np.testing.assert_array_equal(np.concatenate([a, b]), misc["mixup_lambda64"][:8])
    x = np.array([0.5, -1.0, 0.25], dtype=np.float32)
    np.testing.assert_allclose(int16_to_float32(float32_to_int16(x)), x, atol=1 / 32767.)
    create_folder(str(tmp_path / "a" / "b"))
    assert os.path.isdir(str(tmp_path / "a" / "b")) and get_filename("/x/y/main.py") == "main"


def test_move_data_to_device_semantics():
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import move_data_to_device, append_to_dict
    assert move_data_to_device(np.zeros(3, dtype=np.float64), "cpu").dtype == torch.float32
    assert move_data_to_device(np.zeros(3, dtype=np.int16), "cpu").dtype == torch.int64
    names = np.array(["a.wav", "b.wav"])
    assert move_data_to_device(names, "cpu") is names
    d = {}
    append_to_dict(d, "k", 1); append_to_dict(d, "k", 2)
    assert d == {"k": [1, 2]}


def test_config_constants():
    from sound_event_detection_dcase2017_task4_amd.utils import config
    assert (config.sample_rate, config.window_size, config.hop_size, config.mel_bins, config.fmin, config.fmax) == \
        (32000, 1024, 320, 64, 50, 14000)
    assert config.classes_num == 17 and config.frames_per_second == 100 and len(config.ids) == 17
    assert config.lb_to_idx["Train"] == 16 and sum(config.samples_num) == 58662


def test_samplers_and_collate_match_reference_order():
    """TrainSampler reproduces the reference stream incl. its double indexing (data_generator.py:88,:98)."""
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import (TrainSampler, TestSampler, collate_fn,
                                                                               DCASE2017Task4Dataset)
    path = "synthetic:10:3200"
    ts = TrainSampler(path, batch_size=4)
    it = iter(ts)
    got = [m["index_in_hdf5"] for _ in range(4) for m in next(it)]
    rs = np.random.RandomState(1234)
    idx = np.arange(10); rs.shuffle(idx)
    want, pointer = [], 0
    for _ in range(16):
        index = idx[pointer]; pointer += 1
        if pointer >= 10:
            pointer = 0; rs.shuffle(idx)
        want.append(idx[index])
    assert got == want
    batches = list(iter(TestSampler(path, batch_size=4)))
    assert [len(b) for b in batches] == [4, 4, 2] and batches[2][1]["index_in_hdf5"] == 9
    ds = DCASE2017Task4Dataset()
    b = collate_fn([ds[m] for m in batches[0]])
    assert b["waveform"].shape == (4, 3200) and b["waveform"].dtype == np.float32 and b["target"].shape == (4, 17)
    assert b["audio_name"][1] == "syn_00001.wav"
    raw = DCASE2017Task4Dataset(keep_int16=True)[batches[0][0]]["waveform"]
    assert raw.dtype == np.int16 and np.allclose(raw / 32767., b["waveform"][0], atol=1e-7)


def test_cli_flags_match_reference():
    from sound_event_detection_dcase2017_task4_amd.pytorch.main import build_parser
    a = build_parser().parse_args("train --dataset_dir d --workspace w --holdout_fold 1 --model_type Cnn_9layers_FrameAvg "
                                  "--loss_type clip_bce --augmentation mixup --learning_rate 1e-3 --batch_size 32 "
                                  "--resume_iteration 0 --stop_iteration 50000 --cuda".split())
    assert a.mode == "train" and a.batch_size == 32 and a.cuda and a.augmentation == "mixup"
    b = build_parser().parse_args("inference_prob --dataset_dir d --workspace w --holdout_fold 1 --model_type X "
                                  "--loss_type clip_bce --augmentation mixup --batch_size 32 --iteration 50000 --cuda".split())
    assert b.mode == "inference_prob" and b.iteration == 50000


def test_pinned_batch_loader_reproduces_the_dataloader_stream(tmp_path):
    """PinnedBatchLoader (threads + ring of buffers) must hand out exactly the batches of the reference pipeline
    DataLoader(DCASE2017Task4Dataset, batch_sampler=TrainSampler, collate_fn) (data_generator.py:15-164), across the
    sampler's reshuffle at wrap-around, and a batch must stay intact until the next one is requested."""
    import time
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import (DCASE2017Task4Dataset, PinnedBatchLoader,
                                                                                 TestSampler, TrainSampler, collate_fn)
    rs = np.random.RandomState(0)
    N, L = 37, 640
    np.save(tmp_path / "waveform.npy", (rs.randn(N, L) * 3000).astype(np.int16))
    np.save(tmp_path / "target.npy", (rs.rand(N, 17) < 0.2).astype(np.float32))
    np.save(tmp_path / "strong_target.npy", rs.rand(N, 5, 17) < 0.3)
    np.save(tmp_path / "audio_name.npy", np.array([("c%03d.wav" % i).encode() for i in range(N)]))
    root = str(tmp_path)
    ref = torch.utils.data.DataLoader(DCASE2017Task4Dataset(keep_int16=True), batch_sampler=TrainSampler(root, 8),
                                      collate_fn=collate_fn, num_workers=0)
    new = PinnedBatchLoader(root, TrainSampler(root, 8), depth=2, threads=3)
    it1, it2 = iter(ref), iter(new)
    for k in range(12):                                  # 96 draws > 2 epochs of 37 clips
        a, b = next(it1), next(it2)
        if k % 4 == 0:
            time.sleep(0.05)                             # let the producer run ahead as far as it is allowed to
        assert [str(x) for x in a["audio_name"]] == b["audio_name"]
        assert b["waveform"].dtype == torch.int16 and np.array_equal(a["waveform"], b["waveform"].numpy())
        assert np.array_equal(a["target"], b["target"].numpy())
        assert np.array_equal(a["strong_target"], b["strong_target"].numpy())
    it2.close()
    # finite sampler (evaluation order) with a ragged last batch
    got = [b["audio_name"] for b in PinnedBatchLoader(root, TestSampler(root, 8))]
    assert [len(g) for g in got] == [8, 8, 8, 8, 5] and got[0][0] == "c000.wav" and got[-1][-1] == "c036.wav"


def test_parameter_initialisation_follows_the_reference_recipes():
    """SURVEY.md 8a row I1 (reference models.py:15-55): conv / linear / 1x1-conv weights xavier-uniform with zero bias, BN
    gamma = 1 / beta = 0, GRU per-gate uniform(+-sqrt(3/fan_in)) for W_ih and the r, z blocks of W_hh, an orthogonal n
    block, zero biases.  Bit equality with the reference is neither possible nor required (it sets no seed): the
    distributions are checked."""
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    torch.manual_seed(0)
    m = models.Cnn_9layers_Gru_FrameAtt(32000, 1024, 320, 64, 50, 14000, 17)
    sd = m.state_dict()
    for name, cin, cout in (("conv_block1.conv1", 1, 64), ("conv_block2.conv2", 128, 128), ("conv_block4.conv1", 256, 512)):
        w = sd[name + ".weight"]
        bound = (6.0 / (9 * cin + 9 * cout)) ** 0.5
        assert w.abs().max() <= bound * (1 + 1e-6)
        if w.numel() > 10000:                            # uniform: std = bound / sqrt(3), mean 0
            assert abs(w.std().item() / (bound / 3 ** 0.5) - 1) < 0.03 and abs(w.mean().item()) < 0.02 * bound
    for name in ("bn0", "conv_block3.bn2", "att_block.bn_att"):
        assert torch.all(sd[name + ".weight"] == 1) and torch.all(sd[name + ".bias"] == 0)
        assert torch.all(sd[name + ".running_mean"] == 0) and torch.all(sd[name + ".running_var"] == 1)
    for name in ("att_block.att", "att_block.cla"):
        w = sd[name + ".weight"]
        assert w.shape == (17, 512, 1) and w.abs().max() <= (6.0 / (512 + 17)) ** 0.5 * (1 + 1e-6)
        assert torch.all(sd[name + ".bias"] == 0)
    # forward direction: the reference's init_gru recipe (models.py:44-55)
    w_ih, w_hh = sd["gru.weight_ih_l0"], sd["gru.weight_hh_l0"]
    assert w_ih.shape == (768, 512) and w_hh.shape == (768, 256)
    assert w_ih.abs().max() <= (3.0 / 512) ** 0.5 * (1 + 1e-6) and abs(w_ih.std().item() / (1.0 / 512) ** 0.5 - 1) < 0.03
    rz = w_hh[:512]
    assert rz.abs().max() <= (3.0 / 256) ** 0.5 * (1 + 1e-6) and abs(rz.std().item() / (1.0 / 256) ** 0.5 - 1) < 0.03
    n = w_hh[512:].double()
    assert (n @ n.T - torch.eye(256, dtype=torch.float64)).abs().max() < 1e-5
    assert torch.all(sd["gru.bias_ih_l0"] == 0) and torch.all(sd["gru.bias_hh_l0"] == 0)
    # reverse direction: the reference loop never touches `*_reverse`, so nn.GRU's default U(+-1/sqrt(256)) stays -- weights
    # AND (non-zero) biases
    k = 1.0 / 256 ** 0.5
    for name, shape in (("weight_ih_l0_reverse", (768, 512)), ("weight_hh_l0_reverse", (768, 256)),
                        ("bias_ih_l0_reverse", (768,)), ("bias_hh_l0_reverse", (768,))):
        t = sd["gru." + name]
        assert tuple(t.shape) == shape
        assert t.abs().max() <= k * (1 + 1e-6) and abs(t.std().item() / (k / 3 ** 0.5) - 1) < 0.08, name
        assert t.abs().max() > 0.9 * k, name
    nr = sd["gru.weight_hh_l0_reverse"][512:].double()
    assert (nr @ nr.T - torch.eye(256, dtype=torch.float64)).abs().max() > 0.1          # NOT orthogonalised


def test_mel_task_tables_reproduce_the_filter_bank_and_are_bank_conflict_free():
    """ops.mel_task_tables (host side of the log-mel kernel's mel stage): the <= 12-bin task windows with their zero-padded
    weights add up to exactly the 513x64 matrix, every band lists its task slots, and within each group of 32 slots the
    window starts differ mod 32 (the kernel's ds_read_b64 of the (Pa, Pb) pairs is then conflict-free).  Also for filter banks
    other than the reference's (fmin / fmax changed), where the conflict-free matching may fail and any placement is allowed."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.pytorch.models import _slaney_mel
    with pytest.raises(RuntimeError, match="does not fit"):          # a band wider than 4 x 12 bins: refused loudly
        ops.mel_task_tables(np.ascontiguousarray(_slaney_mel(32000, 1024, 64, 0, 16000).T.astype(np.float32)))
    for fmin, fmax, strict in ((50, 14000, True), (300, 8000, False), (20, 12000, False)):
        W = np.ascontiguousarray(_slaney_mel(32000, 1024, 64, fmin, fmax).T.astype(np.float32))     # (513, 64)
        tasks, bands, vals = ops.mel_task_tables(W)
        assert tasks.shape == (128, 4) and bands.shape == (64, 4) and vals.shape == (128 * 12,)
        R = np.zeros_like(W)
        for s in range(128):
            lo, cnt, off, band = tasks[s]
            assert 0 <= lo and lo + 12 <= 513 + 16 and off == 12 * s
            for i in range(12):
                if vals[off + i] != 0:
                    assert cnt == 12 and lo + i < 513
                    R[lo + i, band] += vals[off + i]
        assert np.array_equal(R, W)
        for m in range(64):
            slots = [int(s) for s in bands[m] if s >= 0]
            assert all(tasks[s][3] == m and tasks[s][1] == 12 for s in slots)
            assert len(slots) == len({s for s in range(128) if tasks[s][1] > 0 and tasks[s][3] == m})
        if strict:
            for g in range(4):
                starts = [int(tasks[s][0]) % 32 for s in range(32 * g, 32 * g + 32)]
                assert len(set(starts)) == 32, (g, starts)


def test_loader_refuses_an_experiment_build(tmp_path, monkeypatch):
    """sed_version() carries the hash of the hipcc flags of every object; _lib.verify_flags() refuses a library that was not
    built with build.BASE_FLAGS (a stray SED_HIPCC_FLAGS=-D... once built a wrong-results library that nothing detected)
    unless SED_ALLOW_EXPERIMENT=1.  Built here from heads.hip alone (it holds sed_version) with an extra -D."""
    import ctypes
    import shutil
    import subprocess
    from sound_event_detection_dcase2017_task4_amd import _lib, build
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("needs hipcc")
    _lib._load_hip_runtime()
    libs = {}
    for name, flags in (("exp", build.BASE_FLAGS + ["-DSF_ABL_NOFRAG"]), ("std", list(build.BASE_FLAGS))):
        obj, so = str(tmp_path / (name + ".o")), str(tmp_path / ("lib%s.so" % name))
        subprocess.run(build.compile_cmd(os.path.join(build.CSRC, "heads.hip"), obj, flags), check=True, capture_output=True)
        subprocess.run([os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", so, obj], check=True, capture_output=True)
        libs[name] = ctypes.CDLL(so)
    monkeypatch.delenv("SED_ALLOW_EXPERIMENT", raising=False)
    assert _lib.verify_flags(libs["std"]).endswith("flags:" + build.flags_hash())
    with pytest.raises(RuntimeError, match="experiment build is refused"):
        _lib.verify_flags(libs["exp"])
    monkeypatch.setenv("SED_ALLOW_EXPERIMENT", "1")
    assert "flags:" in _lib.verify_flags(libs["exp"])
    # the shipped library itself was built with the default flags, and no experiment switch is left in the kernels
    assert _lib.lib().sed_version().decode().endswith("flags:" + build.flags_hash())
    src = open(os.path.join(build.CSRC, "conv_sf16.hip")).read()
    assert src.count("#if") <= 3


def test_pinned_batch_loader_hold_keeps_earlier_batches_valid():
    """hold = h: the tensors of a batch stay untouched while h + 1 later batches are requested (the train CLI re-runs the
    batches of refused optimiser steps); hold = 0 is the old contract (valid until the next request)."""
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import PinnedBatchLoader, TestSampler
    path = "synthetic:24:4000"
    for hold in (0, 3):
        loader = PinnedBatchLoader(path, TestSampler(path, batch_size=2), device=None, hold=hold)
        assert loader.depth >= hold + 3
        seen = []
        for batch in loader:
            seen.append((batch["waveform"], batch["waveform"].clone()))
            for wave, copy in seen[-(hold + 1):]:
                assert torch.equal(wave, copy)
        assert len(seen) == 12


def test_weight_cache_trim_keeps_live_entries_of_the_current_generation():
    """ops._WCACHE is trimmed, not cleared: entries of dead tensors / older parameter generations go, live ones of the current
    generation stay -- a prepack call must never wipe what it has just put (CPU tensors suffice: the table only looks at object
    identity, data_ptr and the version counter)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    saved, gen = dict(ops._WCACHE), ops.PARAM_GENERATION
    try:
        ops._WCACHE.clear()
        dead = [torch.zeros(1) for _ in range(70)]
        for t in dead:
            ops._cache_put("k", (t,), 1)
        live = [torch.zeros(1) for _ in range(10)]
        del dead
        for t in live:
            ops._cache_put("k", (t,), 2)
        assert all(ops._cache_fresh("k", (t,)) for t in live)          # the first puts of this batch survived the later ones
        ops.invalidate_weight_caches()
        assert not any(ops._cache_fresh("k", (t,)) for t in live)
        more = [torch.zeros(1) for _ in range(70)]
        for t in more:
            ops._cache_put("k", (t,), 3)
        assert all(ops._cache_fresh("k", (t,)) for t in more) and len(ops._WCACHE) <= 80
    finally:
        ops._WCACHE.clear()
        ops._WCACHE.update(saved)
        ops.PARAM_GENERATION = gen


def test_flags_hash_check_covers_every_object_of_the_library():
    """sed_version() compares the flags hash of EVERY object: the list it walks (SED_OBJECTS in csrc/common.h, plus heads.hip
    itself) must be build.SOURCES, and every source must publish its hash under its own name (round 4: gemm_sf16.o was built
    and linked but missing from the list, so an experiment build of it alone would have gone unnoticed)."""
    from sound_event_detection_dcase2017_task4_amd import build
    common = open(os.path.join(build.CSRC, "common.h")).read()
    listed = re.findall(r"X\((\w+)\)", re.search(r"#define SED_OBJECTS\(X\)(.*)", common).group(1))
    names = [s[:-len(".hip")] for s in build.SOURCES]
    assert sorted(listed + ["heads"]) == sorted(names) and len(set(listed)) == len(listed)
    for n in names:
        src = open(os.path.join(build.CSRC, n + ".hip")).read()
        assert re.search(r"^SED_OBJECT_FLAGS\(%s\)" % n, src, flags=re.M), n
    heads = open(os.path.join(build.CSRC, "heads.hip")).read()
    assert "SED_OBJECTS(SED_WEAK_FLAGS)" in heads and "SED_OBJECTS(SED_FLAGS_ENTRY)" in heads


def test_gpu_tier_collection_order_puts_parity_first_and_spawned_ranks_last():
    """tests/conftest.py orders the `-m gpu` tier (the driver runs it with -x): oracle / golden parity, then the float64 checks
    of the split-f16 kernels, then the rest; CLI subprocesses and multi-rank tests come last, so a process-level failure can
    never hide the numerics evidence again (round 4: test_gpu_parallel sat in front of test_gpu_sf16)."""
    import conftest
    ids = ["tests/test_gpu_parallel.py::a", "tests/test_gpu_cli.py::b", "tests/test_gpu_sf16.py::c", "tests/test_gpu_optim.py::d",
           "tests/test_gpu_model.py::e", "tests/test_gpu_frontend.py::f", "tests/test_gpu_graph.py::g", "tests/test_gpu_ops.py::h",
           "tests/test_gpu_gru.py::i"]
    order = sorted(ids, key=conftest.collection_rank)
    pos = {i.split("::")[1]: k for k, i in enumerate(order)}
    assert max(pos[k] for k in "efhi") < pos["c"] < min(pos["d"], pos["g"]) and max(pos["d"], pos["g"]) < pos["b"] < pos["a"]
    assert order[-1].startswith("tests/test_gpu_parallel.py")


def test_bench_refuses_to_quote_stale_pmc_figures(tmp_path, monkeypatch):
    """bench.py quotes `roofline.traffic` / `mfma_pipe_busy_frac` from committed rocprofv3 PMC digests.  A digest carries the
    hashes of the kernel sources it was collected on (`_meta.sources`); once the sources of the kernel in question differ -- or
    the digest has no record at all -- the figure is NOT quoted (null + the reason), instead of silently describing another
    binary (round-4 review)."""
    import json
    import bench
    from sound_event_detection_dcase2017_task4_amd import build
    cur = build.source_hashes()
    assert set(cur) == set(build.SOURCES) | {"common.h"} and all(len(v) == 12 for v in cur.values())
    assert bench.pmc_fresh({"sources": dict(cur)}, ["conv_sf16_kernel"]) is None
    assert "no kernel-source record" in bench.pmc_fresh(None, ["conv_sf16_kernel"])
    other = dict(cur, **{"conv_sf16.hip": "0" * 12})
    assert "csrc/conv_sf16.hip changed" in bench.pmc_fresh({"sources": other}, ["conv_sf16_kernel"])
    assert bench.pmc_fresh({"sources": other}, ["logmel32_kernel"]) is None            # another kernel's sources: still valid
    for fam, files in bench.KERNEL_SOURCES.items():
        assert all(f in cur for f in files), fam
    # end to end through pmc_traffic(): a fresh digest is quoted, the same digest with a changed source is not
    prof = tmp_path / "profiles" / "r99"
    prof.mkdir(parents=True)
    body = {"void conv_sf16_kernel<4>(Sf16P)": {"launches": 2, "fetch_bytes_raw_per_launch": 50, "fetch_bytes_x2_per_launch": 100,
                                                 "write_bytes_per_launch": 11}}
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    (prof / "pmc_traffic_b32.json").write_text(json.dumps(dict(body, _meta={"sources": cur})))
    val, src = bench.pmc_traffic(["conv_sf16_kernel"], "pmc_traffic_b32.json")
    assert val == 111 and "pmc_traffic_b32.json" in src
    (prof / "pmc_traffic_b32.json").write_text(json.dumps(dict(body, _meta={"sources": other})))
    val, src = bench.pmc_traffic(["conv_sf16_kernel"], "pmc_traffic_b32.json")
    assert val is None and "NOT quoted" in src


def test_hdf5_pack_branch_through_a_stand_in_h5py(tmp_path, monkeypatch):
    """The reference reads its packs with h5py (`h5py.File(path, 'r')`, `hf['waveform'][index]`, `hf.keys()`, context manager:
    data_generator.py:37-47, features.py:232-260).  h5py is not installed here, so the `.h5` branch of `open_store` could never
    run; a stand-in module with exactly that access surface (backed by the arrays of a pack) is put in `sys.modules` and the same
    samplers / dataset / threaded loader must hand out the same batches from `<pack>.h5` as from the `.npy` directory.  (This
    pins the branch's own logic -- names as bytes, int16 rows, the optional strong_target -- not the HDF5 file format.)"""
    import sys
    import types
    from sound_event_detection_dcase2017_task4_amd.utils import data_generator as dg
    rs = np.random.RandomState(3)
    N, L = 21, 320
    arrays = {"waveform": (rs.randn(N, L) * 2000).astype(np.int16), "target": (rs.rand(N, 17) < 0.2).astype(np.float32),
              "strong_target": rs.rand(N, 3, 17) < 0.3, "audio_name": np.array([("h%02d.wav" % i).encode() for i in range(N)])}
    d = tmp_path / "pack"
    d.mkdir()
    for k, v in arrays.items():
        np.save(d / (k + ".npy"), v)
    opened = []

    class File(object):                                   # the h5py.File surface the reference (and this build) uses
        def __init__(self, path, mode):
            assert mode == "r" and path.endswith(".h5")
            opened.append(path)
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def __getitem__(self, k):
            return arrays[k]
        def keys(self):
            return arrays.keys()
        def close(self):
            pass
    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    h5 = str(tmp_path / "training.h5")
    open(h5, "wb").close()                                # a FILE ending in .h5 selects the h5py branch
    a = dg.PinnedBatchLoader(h5, dg.TrainSampler(h5, 6), depth=2, threads=2)
    b = dg.PinnedBatchLoader(str(d), dg.TrainSampler(str(d), 6), depth=2, threads=2)
    ia, ib = iter(a), iter(b)
    for _ in range(9):                                    # > 2 epochs: across the sampler's reshuffle
        x, y = next(ia), next(ib)
        assert x["audio_name"] == y["audio_name"] and x["waveform"].dtype == torch.int16
        for k in ("waveform", "target", "strong_target"):
            assert torch.equal(x[k], y[k]), k
    ia.close(); ib.close()
    item = dg.DCASE2017Task4Dataset()[{"hdf5_path": h5, "index_in_hdf5": 4}]
    assert item["audio_name"] == "h04.wav" and item["waveform"].dtype == np.float32
    np.testing.assert_allclose(item["waveform"], arrays["waveform"][4] / 32767.0, rtol=1e-6)
    assert opened


def test_reference_optimizer_state_and_streams_resume_on_the_host(tmp_path):
    """Host side of resume / checkpoint interchange (reference main.py:125-134, :144-145, :221-231), no GPU needed:
    (a) optim.torch_adam_state_to_flat turns a stock torch.optim.Adam(amsgrad=True).state_dict() over model.parameters() -- frozen
        tensors occupy indices, never-used parameters carry no state -- into flat moments over the trainable parameters, and
        refuses anything else with the reason;
    (b) TrainSampler.skip(n) leaves the seed-1234 stream exactly where n iterated batches leave it (wrap-around reshuffles included);
    (c) the mixup generator's state survives a checkpoint as plain tensors / scalars (torch.load with weights_only=True)."""
    import io
    import torch
    from sound_event_detection_dcase2017_task4_amd.optim import torch_adam_state_to_flat
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import TrainSampler
    from sound_event_detection_dcase2017_task4_amd.utils.utilities import Mixup, random_state_from_plain, random_state_to_plain
    torch.manual_seed(0)
    params = [torch.randn(5, 3), torch.randn(4, requires_grad=True), torch.randn(2, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    opt = torch.optim.Adam(params, lr=2e-3, betas=(0.8, 0.95), eps=1e-7, weight_decay=0., amsgrad=True)
    for _ in range(3):
        params[1].grad = torch.randn(4); params[2].grad = torch.randn(2, 3); params[3].grad = None       # params[3]: never used
        opt.step()
    numels, trainable = [p.numel() for p in params], [p.requires_grad for p in params]
    step, m, v, vmax, hyper = torch_adam_state_to_flat(opt.state_dict(), numels, trainable)
    assert step == 3 and hyper == {"lr": 2e-3, "betas": (0.8, 0.95), "eps": 1e-7} and m.numel() == 4 + 6 + 7
    st = opt.state_dict()["state"]
    assert torch.equal(m[:4], st[1]["exp_avg"]) and torch.equal(v[4:10], st[2]["exp_avg_sq"].reshape(-1))
    assert torch.equal(vmax[:4], st[1]["max_exp_avg_sq"]) and float(m[10:].abs().sum()) == 0 and float(vmax[10:].abs().sum()) == 0
    with pytest.raises(ValueError, match="WITHOUT amsgrad"):
        torch_adam_state_to_flat(torch.optim.Adam(params, lr=1e-3).state_dict(), numels, trainable)
    with pytest.raises(ValueError, match="holds 4 parameters, this model has 3"):
        torch_adam_state_to_flat(opt.state_dict(), numels[:3], trainable[:3])
    with pytest.raises(ValueError, match="weight_decay"):
        torch_adam_state_to_flat(torch.optim.Adam(params, lr=1e-3, amsgrad=True, weight_decay=0.1).state_dict(), numels, trainable)
    with pytest.raises(ValueError, match="has 4 elements, the model's parameter 9"):
        torch_adam_state_to_flat(opt.state_dict(), [15, 9, 6, 7], trainable)
    with pytest.raises(ValueError, match="frozen in this model"):
        torch_adam_state_to_flat(opt.state_dict(), numels, [False, False, True, True])
    with pytest.raises(ValueError, match="ONE parameter group"):
        torch_adam_state_to_flat({"state": {}, "param_groups": []}, numels, trainable)
    # (b) sampler: 23 clips, batches of 8 -> wraps (and reshuffles) every third batch
    for n in (0, 1, 2, 3, 7, 40):
        a, b = TrainSampler("synthetic:23:1000", 8), TrainSampler("synthetic:23:1000", 8)
        it = iter(a)
        for _ in range(n):
            next(it)
        b.skip(n)
        ib = iter(b)
        for _ in range(4):
            assert [x["index_in_hdf5"] for x in next(it)] == [x["index_in_hdf5"] for x in next(ib)], n
    # (c) mixup stream through a checkpoint file
    mix = Mixup(1.)
    for _ in range(5):
        mix.get_lambda(64)
    path = str(tmp_path / "ck.pth")
    torch.save({"streams": {"mixup_rng": random_state_to_plain(mix.random_state), "torch_rng": torch.get_rng_state()}}, path)
    ck = torch.load(path, map_location="cpu")                     # default weights_only=True must accept it
    mix2 = Mixup(1., random_seed=99)
    random_state_from_plain(mix2.random_state, ck["streams"]["mixup_rng"])
    assert np.array_equal(mix.get_lambda(64), mix2.get_lambda(64))


def test_unsupported_constructor_arguments_name_the_supported_set():
    """reference models.py:238-262 takes any (window_size, hop_size, mel_bins); this build's kernels are specialised for the
    config.py values (and the reference's own bn0 = BatchNorm2d(64) pins mel_bins too).  Anything else is refused at construction
    with a ValueError -- an Exception, the reference's error style -- that names the supported set and what was passed; the
    free arguments (sample_rate, fmin, fmax, classes_num) construct."""
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    for bad in ((32000, 2048, 320, 64, 50, 14000, 17), (32000, 1024, 160, 64, 50, 14000, 17), (32000, 1024, 320, 128, 50, 14000, 17)):
        with pytest.raises(ValueError, match="window_size=1024, hop_size=320, mel_bins=64") as e:
            models.Cnn_9layers_FrameAvg(*bad)
        assert "Incorrect argument!" in str(e.value) and isinstance(e.value, Exception)
        assert ("window_size=%r, hop_size=%r, mel_bins=%r" % bad[1:4]) in str(e.value)
    m = models.Cnn_9layers_FrameAvg(16000, 1024, 320, 64, 20, 7000, 5)          # the free ones
    assert m.fc.weight.shape == (5, 512) and m.logmel_extractor.melW.shape == (513, 64)
    with pytest.raises(ValueError, match=r"supports \(8, 512, 64, 64\) only"):
        models.MultiHead(4, 512, 128, 128)
