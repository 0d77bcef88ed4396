#!/usr/bin/env python
"""Headline benchmark: training clips/s (10 s @ 32 kHz) of Cnn_9layers_FrameAvg bs=32 on MI355X, synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: log-mel -> bn0+SpecAugment+mixup -> 4 ConvBlocks -> head ->
clip_bce -> backward (the RCCL all-reduce of the flat gradient runs in buckets beside it) -> Adam-amsgrad, with the
waveforms already resident in HBM.  Workload = the configuration BASELINE.json's metric is quoted on: Cnn_9layers_FrameAvg,
batch_size 32 (post-mixup clips per GPU = 64 waveforms per step, mixup + SpecAugment on: the reference README's training
command); weak scaling (batch per GPU fixed).  `--gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment)
re-executes itself as N ranks under torch.distributed.run and refuses to run if the node has fewer than N GPUs.  Prints ONE
JSON line on rank 0: `value` / `ms_per_step` from a timed region with NO instrumentation inside it; `roofline` (the dominant kernel
family, HIP-event timed live over the same K steps run again straight behind it: `roofline_pass`), `kernels` (every MFMA family,
from a third pass with the weight gradients on the main stream so that no duration includes waiting beside another kernel),
`cpu_baseline` (the CPU oracle timed on this host's cores, config 0) and, at N=1, `extra_configs`: BASELINE.json configs[1]
(B=256, with its own roofline / front-end / traffic objects) first, then the other configurations, a few steps each in the same
process, and last the two input-path rows of SURVEY.md 8(d) (H2D-inclusive; the train CLI's loader in the loop).
"""
import argparse
import gc
import glob
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from sound_event_detection_dcase2017_task4_amd import ops, parallel
from sound_event_detection_dcase2017_task4_amd.graph import GraphCaptureError, GraphedTrainStep
from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
from sound_event_detection_dcase2017_task4_amd.pytorch import models
from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func
from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
from sound_event_detection_dcase2017_task4_amd.utils.utilities import Mixup

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16), the pipe the split-f16 kernels run on
# What a BARE stream of that MFMA sustains on these boxes with RANDOM operands (tools/mfma_f16_ubench.hip, no memory traffic at
# all; profiles/r03/mfma_f16_ubench.txt): the part clocks to its power budget and switching activity is data dependent --
# 2.04-2.17 PF with smooth operands, 1.62-1.64 PF with random mantissas (what hi / lo halves of real activations are).
F16_MFMA_SUSTAINED_RANDOM_TFLOPS = 1630.0
HBM_PEAK_GBPS = 8000.0
CTOR = (32000, 1024, 320, 64, 50, 14000, 17)


def synth_batch(B2, L, seed, device):
    """SURVEY.md §8d synthetic inputs: N(0, 0.1^2) clipped to [-1, 1]; Bernoulli(0.2) targets."""
    g = torch.Generator(device=device).manual_seed(seed)
    wave = (torch.randn((B2, L), generator=g, device=device, dtype=torch.float32) * 0.1).clamp_(-1.0, 1.0)
    target = (torch.rand((B2, 17), generator=g, device=device) < 0.2).float()
    return wave, target


def cpu_baseline(batch=32, clips=128, seconds=10, threads=0):
    """Config 0 on the host cores with the CPU oracle (a "port": the reference's Python cannot travel):
    Cnn_9layers_FrameAvg, B=32, clip_bce, no mixup (SpecAugment on), Adam-amsgrad, `clips`/32 steps; first step =
    warm-up, the remaining three timed.  (The reference's own code re-measured in the build container:
    tools/ref_cpu_baseline.py -> profiles/.)"""
    from oracle import model as om
    threads = threads or min(os.cpu_count() or 1, 32)      # torch CPU conv scaling flattens/regresses beyond ~32 threads
    torch.set_num_threads(threads)
    mt = "Cnn_9layers_FrameAvg"
    st = om.recipe_state(mt, 2)
    keys = om.trainable_keys(mt)
    for k in keys:
        st[k].requires_grad_(True)
    opt_state = {k: [torch.zeros_like(st[k]) for _ in range(3)] for k in keys}
    L = 32000 * seconds
    rs = np.random.RandomState(1234)
    times = []
    for it in range(clips // batch):
        x = torch.from_numpy((rs.randn(batch, L) * 0.1).astype(np.float32))
        y = torch.from_numpy((rs.rand(batch, 17) < 0.2).astype(np.float32))
        t0 = time.time()
        o = om.forward(mt, st, x, training=True)
        loss = om.clip_bce(o, {"target": y})
        grads = torch.autograd.grad(loss, [st[k] for k in keys])
        with torch.no_grad():
            for k, g in zip(keys, grads):
                m, v, vm = opt_state[k]
                om.adam_amsgrad_step(st[k], g, m, v, vm, it + 1, 1e-3)
        times.append(time.time() - t0)
    timed = times[1:] if len(times) > 1 else times
    return {"value": round(batch * len(timed) / sum(timed), 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "config 0: %d x %d-clip train steps (10 s clips, no mixup, SpecAugment on, Adam-amsgrad) of the "
                      "CPU oracle (torch fp32, %d threads); first step warm-up, %d timed (%s s)"
                      % (len(times), batch, threads, len(timed), "/".join("%.1f" % t for t in timed))}


def latest_profile(name):
    """Newest profiles/rNN/<name> (the PMC digests are produced per round by tools/pmc_digest.py)."""
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9]*", name)))
    return cands[-1] if cands else None


# which kernel sources a kernel family is compiled from: a committed PMC figure is quoted only while THESE files are the ones it
# was collected on (`_meta.sources` of the digest, written on the GPU box at collection time by tools/collect_profiles.sh)
KERNEL_SOURCES = {"conv_sf16_kernel": ["conv_sf16.hip", "common.h"], "wgrad_sf16_": ["conv_sf16.hip", "common.h"],
                  "logmel32_kernel": ["logmel.hip", "common.h"], "conv_wino2_kernel": ["conv_wino2.hip", "common.h"],
                  "wgrad_wino2_": ["conv_wino2.hip", "common.h"], "conv_igemm_kernel": ["conv.hip", "common.h"],
                  "wgrad_kernel": ["conv.hip", "common.h"], "gemm_sf16_kernel": ["gemm_sf16.hip", "common.h"],
                  "gemm_tn_sf16_": ["gemm_sf16.hip", "common.h"]}


def pmc_fresh(meta, substrings, current=None):
    """None when the digest's recorded kernel sources equal the current ones for every kernel in `substrings`, else the reason
    it must not be quoted (no record at all counts as stale)."""
    if current is None:
        from sound_event_detection_dcase2017_task4_amd import build
        current = build.source_hashes()
    rec = (meta or {}).get("sources")
    if not rec:
        return "no kernel-source record (`_meta.sources`) in the digest"
    for sub in substrings:
        for f in KERNEL_SOURCES.get(sub, sorted(current)):
            if rec.get(f) != current.get(f):
                return "csrc/%s changed since the counters were collected (%s then, %s now)" % (f, rec.get(f), current.get(f))
    return None


def pmc_traffic(substrings, name="pmc_traffic.json"):
    """HBM bytes per launch of a timed kernel family, from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE; tools/pmc_digest.py).  PMC collection needs rocprofv3 around the process, so it cannot
    be sampled live: the figure is valid for the workload the file was collected on (pmc_traffic_b32.json: the headline
    bs=32; pmc_traffic.json: configs[1], B=256) AND for the kernel sources it was collected on (pmc_fresh), otherwise null
    with the reason as the source."""
    path = latest_profile(name)
    if not substrings or path is None:
        return None, None
    data = json.load(open(path))
    stale = pmc_fresh(data.get("_meta"), substrings)
    if stale:
        return None, "%s NOT quoted: %s" % (os.path.relpath(path, REPO), stale)
    tot, n = 0.0, 0
    for k, v in data.items():
        if k != "_meta" and any(sub in k for sub in substrings):
            tot += (v["fetch_bytes_x2_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"] if "reduce" not in k else 0
    return (round(tot / n) if n else None), "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes/launch)" % os.path.relpath(path, REPO)


FAMILY_KERNELS = {"gemm_nt_sf16_mfma": ["gemm_sf16_kernel"], "gemm_tn_sf16_mfma(+slice reduce)": ["gemm_tn_sf16_"], "conv3x3_sf16_mfma(fwd+dgrad)": ["conv_sf16_kernel"], "conv3x3_wgrad_sf16_mfma(+slice reduce)": ["wgrad_sf16_"], "conv3x3_wino2d_mfma(fwd+dgrad)": ["conv_wino2_kernel"],
                  "conv3x3_wgrad_wino2d_mfma(+slice reduce)": ["wgrad_wino2_"],
                  "conv3x3_igemm_mfma(fwd+dgrad)": ["conv_igemm_kernel"], "conv3x3_wgrad_mfma(+slice reduce)": ["wgrad_kernel"]}


NOTES = {"conv3x3_wino2d_mfma(fwd+dgrad)": ("fused 2-D Winograd F(2x2,3x3) implicit GEMM on fp32 MFMA", 1 / 2.25, FP32_MFMA_PEAK_TFLOPS),
         "conv3x3_wgrad_wino2d_mfma(+slice reduce)": ("Winograd-domain F(2x2,3x3) weight gradient on fp32 MFMA", 1 / 2.25, FP32_MFMA_PEAK_TFLOPS),
         "conv3x3_wgrad_sf16_mfma(+slice reduce)": ("weight gradient with split-f16 operands (LDS transpose reads) on the f16 MFMA pipe",
                                                    3.0, F16_MFMA_PEAK_TFLOPS),
         "gemm_tn_sf16_mfma(+slice reduce)": ("dense TN GEMM with split-f16 operands (weight gradients of the GRU / MultiHead layers) on the f16 MFMA pipe", 3.0, F16_MFMA_PEAK_TFLOPS),
         "gemm_nt_sf16_mfma": ("dense NT GEMM with split-f16 operands (GRU / MultiHead projections) on the f16 MFMA pipe", 3.0, F16_MFMA_PEAK_TFLOPS),
         "conv3x3_sf16_mfma(fwd+dgrad)": ("direct 3x3 convolution with split-f16 operands (hi*hi + hi*lo + lo*hi, exact products, fp32 "
                                          "accumulation: the error of a direct fp32 convolution) on the f16 MFMA pipe", 3.0, F16_MFMA_PEAK_TFLOPS)}


def kernel_report(timing, steps, B2, default_workload, by_shape=False, frames=1001, pmc_file="pmc_traffic.json"):
    """(per-family MFMA kernel table, `roofline` of the dominant family, `roofline_frontend`) from the HIP events ops.TIMING
    collected inside a timed region.  'achieved' always counts the ALGORITHMIC direct-convolution flops:
    Winograd kernels execute fewer MACs than that on the fp32 MFMA pipe, the split-f16 kernels THREE f16 MACs per
    algorithmic MAC on the f16 MFMA pipe (`executed_*`)."""
    busy_file = "pmc_mfma_busy_b32.json" if pmc_file == "pmc_traffic_b32.json" else "pmc_mfma_busy.json"
    timing = dict(timing or {})

    def summarise(groups):
        out = {}
        for tag, evs in groups.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            fl = sum(f for _, _, f in evs)
            out[tag] = {"launches": len(evs), "ms_total": round(ms, 3), "avg_ms": round(ms / max(len(evs), 1), 4),
                        "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None}
        return out

    fam = {}
    fe = timing.pop("logmel_frontend", None)
    for tag, evs in timing.items():
        fam.setdefault(tag.split("|")[0], []).extend(evs)
    kern = summarise(fam)
    frontend = None
    if fe:
        ms = sum(a.elapsed_time(b) for a, b, _ in fe)
        gbps = sum(nb for _, _, nb in fe) / (ms * 1e-3) / 1e9
        tr, src = pmc_traffic(["logmel32_kernel"], pmc_file) if default_workload else (None, None)
        # what bounds it: the FFT runs on the vector ALU + LDS (4450 cycles per frame pair against a
        # VALU-only floor of 2300 = 0.25 ms per 512 waveforms, tools/valu_ubench.hip); HBM traffic is 1.06x algorithmic
        valu_floor_ms = 0.25 * (B2 * frames) / (512.0 * 1001.0)
        frontend = {"kernel": "logmel32_kernel (STFT+mel+log, K1)", "bound": "valu", "achieved": round(gbps, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                    "target_frac": 0.6, "met": bool(gbps / HBM_PEAK_GBPS >= 0.6), "traffic": tr,
                    "traffic_source": src, "avg_launch_ms": round(ms / len(fe), 4),
                    "bytes_per_waveform": int(fe[0][2] / B2),
                    "valu_floor_ms": round(valu_floor_ms, 4), "frac_of_valu_floor": round(valu_floor_ms / (ms / len(fe)), 4),
                    "note": "achieved / peak / frac are the HBM figures north_star asks for (algorithmic bytes / time vs 8 TB/s); "
                            "the kernel is bound by the vector ALU + LDS of its radix-32 FFT, not by HBM (PMC traffic = 1.06x "
                            "algorithmic): frac_of_valu_floor = the instruction-count floor of that FFT / measured time"}
    if by_shape:
        for tag, v in sorted(summarise(timing).items()):
            print("# %-62s %3d launches  %8.3f ms/launch  %6.1f TFLOP/s" % (tag, v["launches"], v["avg_ms"], v["tflops"]),
                  file=sys.stderr)
    dom = max(kern, key=lambda k: kern[k]["ms_total"]) if kern else None
    roofline = None
    if dom is not None:
        what, executed_per_alg, peak = NOTES.get(dom, ("fp32 MFMA implicit GEMM", 1.0, FP32_MFMA_PEAK_TFLOPS))
        roofline = {"kernel": dom, "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": peak,
                    "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / peak, 4), "traffic": None,
                    "launches_per_step": kern[dom]["launches"] // steps, "avg_launch_ms": kern[dom]["avg_ms"]}
        roofline["note"] = ("%s: 'achieved' counts the ALGORITHMIC direct-convolution flops (SURVEY.md 8d), 'peak' is the dense "
                            "peak of the MFMA pipe the kernel runs on; the kernel executes %.4gx the algorithmic flops there"
                            % (what, executed_per_alg))
        roofline["executed_tflops"] = round(kern[dom]["tflops"] * executed_per_alg, 2)
        roofline["executed_frac"] = round(kern[dom]["tflops"] * executed_per_alg / peak, 4)
        if peak == F16_MFMA_PEAK_TFLOPS:
            roofline["sustained_peak_random_operands"] = F16_MFMA_SUSTAINED_RANDOM_TFLOPS
            roofline["executed_frac_of_sustained"] = round(kern[dom]["tflops"] * executed_per_alg / F16_MFMA_SUSTAINED_RANDOM_TFLOPS, 4)
            roofline["sustained_note"] = ("a bare v_mfma_f32_32x32x16_f16 stream with random operands sustains 1.62-1.64 PF on this part "
                                          "(power-limited clock; 2.04-2.17 PF with smooth operands): profiles/r03/mfma_f16_ubench.txt, "
                                          "tools/mfma_f16_ubench.hip -- `frac` / `executed_frac` stay relative to the 2.5 PF datasheet peak")
        if default_workload:
            roofline["traffic"], src = pmc_traffic(FAMILY_KERNELS.get(dom), pmc_file)
            if src:
                roofline["traffic_source"] = src
        busy = latest_profile(busy_file) if default_workload else None
        if busy:                                    # committed PMC evidence: share of GPU cycles the MFMA pipe is executing
            data = json.load(open(busy))
            stale = pmc_fresh(data.get("_meta"), FAMILY_KERNELS.get(dom, []))
            if stale:
                roofline["mfma_pipe_busy_frac"] = None
                roofline["mfma_pipe_busy_source"] = "%s NOT quoted: %s" % (os.path.relpath(busy, REPO), stale)
            else:
                for fam, v in data.items():
                    if fam != "_meta" and any(fam.startswith(x.rstrip("_")) or x.rstrip("_") in fam for x in FAMILY_KERNELS.get(dom, [])):
                        roofline["mfma_pipe_busy_frac"] = [v["mfma_busy_frac_min"], v["mfma_busy_frac_max"]]
                        roofline["mfma_pipe_busy_source"] = "%s (%s)" % (os.path.relpath(busy, REPO), v["definition"])
    # every MFMA kernel family of the step, same accounting (the dominant one above is the `roofline` object)
    for tag, v in kern.items():
        what, executed_per_alg, peak = NOTES.get(tag, ("fp32 MFMA implicit GEMM", 1.0, FP32_MFMA_PEAK_TFLOPS))
        if v["tflops"]:
            v["executed_frac_of_pipe_peak"] = round(v["tflops"] * executed_per_alg / peak, 4)
    return kern, roofline, frontend


class Workload(object):
    """One configuration of the hot path on this rank: model + optimiser + a resident pool of synthetic batches."""

    def __init__(self, model_type, B, mix, rank, world, dev, seconds=10, inference=False, int16=False, h2d=False, hip_graph=False,
                 loader_clips=0):
        self.mt, self.B, self.mix, self.inference, self.h2d = model_type, B, mix, inference, h2d
        self.graphed, self.hip_graph_error = None, None
        self.loader = self.loader_iter = None
        self.rank, self.world, self.dev = rank, world, dev
        self.B2 = 2 * B if (mix and not inference) else B
        L = 32000 * seconds
        torch.manual_seed(1234 + rank)
        self.model = getattr(models, model_type)(*CTOR).to(dev)
        self.model.train()
        self.opt = FusedAdamAmsgrad(self.model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, world_size=world)
        parallel.broadcast_flat(self.opt.flat)
        parallel.broadcast_buffers(self.model)
        self.loss_func = get_loss_func("clip_bce")
        self.mixup = Mixup(mixup_alpha=1., random_seed=1234 + rank)
        self.pool = [synth_batch(self.B2, L, 1000 * rank + i, dev) for i in range(2)] if not loader_clips else []
        if loader_clips:
            # the train CLI's input pipeline in the loop (pytorch/main.py: seed-1234 TrainSampler -> PinnedBatchLoader: threads fill
            # page-locked int16 buffers from the packed store, uploads run one batch ahead on a copy stream) on an in-memory
            # synthetic store -- what reference main.py:160-170, :238-239 do with DataLoader workers + move_data_to_device
            from sound_event_detection_dcase2017_task4_amd.utils.data_generator import PinnedBatchLoader, TrainSampler
            path = "synthetic:%d:%d" % (loader_clips, L)
            self.loader = PinnedBatchLoader(path, TrainSampler(path, self.B2, random_seed=1234 + rank), device=dev)
            self.loader_iter = iter(self.loader)
        if int16 or h2d:
            self.pool = [((w * 32767.0).round().to(torch.int16), t) for (w, t) in self.pool]
        if h2d:                       # double-buffered upload on a copy stream, one batch ahead of the compute
            self.host_pool = [w.cpu().pin_memory() for (w, _) in self.pool]
            self.copy_stream = torch.cuda.Stream()
            self.dbuf = [torch.empty_like(self.pool[0][0]) for _ in range(2)]
            self.dev_ready = [torch.cuda.Event() for _ in range(2)]
            self.upload(0)
        if inference:
            self.model.eval()
        elif hip_graph and not h2d:
            # forward + loss + backward replayed as ONE hipGraphLaunch after `hip_graph` eager steps (graph.GraphedTrainStep); the
            # batch is copied into the graph's static input buffers every step (inside the timed region)
            self.graphed = GraphedTrainStep(self.model, self.opt, self.loss_func, mixup=mix, eager_steps=hip_graph)

    def upload(self, i):
        self.copy_stream.wait_stream(torch.cuda.current_stream())     # the buffer's previous reader (step i-2) is done
        with torch.cuda.stream(self.copy_stream):
            self.dbuf[i % 2].copy_(self.host_pool[i % len(self.host_pool)], non_blocking=True)
            self.dev_ready[i % 2].record(self.copy_stream)

    def step(self, i):
        if self.loader_iter is not None:
            batch = next(self.loader_iter)
            wave, target = batch["waveform"], batch["target"]
        else:
            wave, target = self.pool[i % len(self.pool)]
        if self.h2d:
            torch.cuda.current_stream().wait_event(self.dev_ready[i % 2])
            wave = self.dbuf[i % 2]
            self.upload(i + 1)
        if self.inference:
            with torch.no_grad():
                out = self.model(wave, None)
            return out["clipwise_output"].sum()
        if self.graphed is not None:
            lam_h = self.mixup.get_lambda(self.B2) if self.mix else None
            try:
                return self.graphed(wave, target, lam_h)
            except GraphCaptureError as e:     # capture refused (on every rank together; nothing of this step has run): eager from here on
                self.hip_graph_error, self.graphed = repr(e), None
                torch.cuda.synchronize()
                if self.mix:                   # the lambdas drawn for this step are used by the eager body below
                    lam = ops.upload_small(lam_h, self.dev, torch.float32)
                    out = self.model(wave, lam)
                    loss = self.loss_func(out, {"target": do_mixup(target, lam)})
                    self.opt.zero_grad()
                    loss.backward()
                    self.opt.step()
                    return loss
        if self.mix:
            lam = ops.upload_small(self.mixup.get_lambda(self.B2), self.dev, torch.float32)   # pinned staging, async
            out = self.model(wave, lam)
            tgt = do_mixup(target, lam)
        else:
            out = self.model(wave, None)
            tgt = target
        loss = self.loss_func(out, {"target": tgt})
        self.opt.zero_grad()
        loss.backward()                  # gradient buckets go to RCCL as they complete (parallel.GradBuckets)
        self.opt.step()                  # waits for the buckets, then ONE Adam kernel over the flat buffer (1/world folded in)
        return loss

    def graph_info(self):
        if self.graphed is None:
            return {"hip_graph": False, "hip_graph_error": self.hip_graph_error} if self.hip_graph_error else {"hip_graph": False}
        return {"hip_graph": self.graphed.replays > 0, "hip_graph_replays": self.graphed.replays,
                "hip_graph_note": "forward + loss + backward = one hipGraphLaunch per step (graph.GraphedTrainStep); inputs are "
                                  "copied into its static buffers and the Adam kernel is launched behind it, both inside the timed region"}

    def sync(self):
        if self.world > 1 or parallel.collectives_on():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup, timing=False, timing_only=None):
        """W untimed steps, then exactly K timed ones bracketed by barrier + synchronize; MAX over ranks."""
        ev0 = len(self.opt.buckets.wait_events) if self.opt.buckets.wait_events is not None else 0
        for i in range(warmup):
            self.step(i)
        self.sync()
        if getattr(self, "first_run_steps", None) is None:        # where the timed steps of the FIRST run sit in wait_events
            self.first_run_steps = (ev0 + warmup, steps)
        if timing:
            ops.TIMING, ops.TIMING_ONLY = {}, timing_only
        t0 = time.time()
        for i in range(steps):
            loss = self.step(warmup + i)
        self.sync()
        dt = time.time() - t0
        tm, ops.TIMING, ops.TIMING_ONLY = ops.TIMING, None, None
        ops.check_device_errors()
        if self.world > 1 or parallel.collectives_on():
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, float(loss.item()), tm

    def host_enqueue_ms(self, steps=5):
        """Host time to ENQUEUE one step, un-throttled: `steps` <= 5 steps issued from a drained device -- fewer than the 8
        slots of the pinned upload ring (ops.upload_small waits for the copy of 8 uploads ago) and nothing else blocks, so
        the host never waits for the GPU inside the bracket.  (Round 3 reported the average over the whole timed region,
        where the ring throttles the host to the GPU's pace: that number followed the step count, not the launch cost.)"""
        steps = min(int(steps), 5)
        self.sync()
        t0 = time.time()
        for i in range(steps):
            self.step(i)
        ms = (time.time() - t0) / steps * 1e3
        self.sync()
        return round(ms, 3)

    def kernel_table(self, steps, default_workload, pmc_file, by_shape=False, frames=1001):
        """Every MFMA kernel family from a pass with ALL kernels on one stream (weight gradients not on the side stream): a
        duration measured beside another kernel includes the time spent waiting for CU slots (round 3's driver line
        showed 24.4 ms of weight gradients where the one-stream profile has 17.3).  Returns (kernels, roofline, frontend,
        ms_per_step of that pass)."""
        prev, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, False
        try:
            dt, _, tm = self.run(steps, 1, timing=True)
        finally:
            ops.WGRAD_SIDE_STREAM = prev
        kern, roof, fe = kernel_report(tm, steps, self.B2, default_workload, by_shape=by_shape, frames=frames, pmc_file=pmc_file)
        return kern, roof, fe, dt / steps * 1e3

    def close(self):
        if self.loader_iter is not None:
            self.loader_iter.close()
            self.loader_iter = None

    def describe(self, seconds=10, int16=False):
        where = ("waveforms resident in HBM" if not (self.h2d or self.loader) else
                 "waveforms uploaded from pinned host memory one batch ahead on a copy stream (PCIe inside the step)" if self.h2d else
                 "waveforms through the train CLI's input pipeline (sampler -> PinnedBatchLoader threads -> pinned int16 -> H2D copy stream)")
        return "%s, batch_size=%d per GPU%s, %d s @ 32 kHz %s %s, %s" % (
            self.mt, self.B, (" (post-mixup clips), mixup (%d waveforms/step/GPU)" % self.B2) if (self.mix and not self.inference)
            else ", no mixup", seconds, "int16" if (int16 or self.h2d or self.loader) else "fp32", where,
            "eval-mode forward only" if self.inference else "SpecAugment on, clip_bce, Adam-amsgrad")


def graph_eager_steps(mode, B, world, warmup, inference=False, h2d=False):
    """Eager steps before the HIP-graph capture (0 = no graph).  The capture must fall inside the warm-up (it takes a few
    hundred ms) behind at least one eager step (lazily built tables)."""
    on = mode == "on"
    if not on or world > 1 or inference or h2d or warmup < 2:
        return 0
    return min(3, warmup - 1)


DOMINANT = ("conv3x3_sf16_mfma(fwd+dgrad)", "conv3x3_wino2d_mfma(fwd+dgrad)",
            "conv3x3_igemm_mfma(fwd+dgrad)", "logmel_frontend")     # families bracketed by HIP events inside the headline region


def measure(w, steps, warmup, default_workload, pmc_file, by_shape=False, frames=1001, events_in_region=True):
    """The full set of numbers of one training workload: (row dict).  `value` / `ms_per_step` come from a timed region on the
    default schedule with NO instrumentation inside it (round 5's headline carried 15 HIP-event pairs per step: +0.15 ms).  The
    `roofline` / `roofline_frontend` objects are measured live over the SAME K steps run again straight behind it, with HIP-event
    pairs around the forward / dgrad convolutions and the log-mel kernel on the stream they are launched on (those kernels never run
    beside another one, so the side-stream weight gradients do not distort them; its own ms is reported beside the headline's);
    `kernels` from a third, one-stream pass."""
    dt, loss, _ = w.run(steps, warmup)
    row = {"value": round(w.B * w.world * steps / dt, 2), "unit": "clips/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "loss": round(loss, 5), "roofline": None, "roofline_frontend": None}
    row.update(w.graph_info())
    row["ms_per_step_without_kernel_events"] = row["ms_per_step"]          # (the name round 5 reported this figure under)
    row["value_without_kernel_events"] = row["value"]
    if events_in_region:
        dte, _, tm = w.run(steps, 1, timing=True, timing_only=DOMINANT)
        _, roof, fe = kernel_report(tm, steps, w.B2, default_workload, frames=frames, pmc_file=pmc_file)
        row["roofline"], row["roofline_frontend"] = roof, fe
        row["roofline_pass"] = {"ms_per_step": round(dte / steps * 1e3, 3), "steps": steps,
                                "note": "the headline's K steps run again on the same schedule with HIP-event pairs around the "
                                        "forward / dgrad convolutions and the log-mel kernel (an event pair costs the stream ~6 us): "
                                        "`roofline` / `roofline_frontend` are measured over THIS region, `value` over the event-free one"}
    if w.world == 1:
        ksteps = max(2, min(steps, 10))
        kern, roof2, _, ms2 = w.kernel_table(ksteps, default_workload, pmc_file, by_shape=by_shape, frames=frames)
        row["kernels"] = kern
        row["kernels_pass"] = {"ms_per_step": round(ms2, 3), "steps": ksteps,
                               "note": "every MFMA family HIP-event-timed with the weight gradients on the MAIN stream (no "
                                       "kernel runs beside another: clean durations); an event pair costs the stream ~6 us"}
        row["mfma_kernels_share_of_step"] = round(sum(v["ms_total"] for v in kern.values()) / (ms2 * ksteps), 4)
        row["host_enqueue_ms_per_step"] = w.host_enqueue_ms()
    return row, dt


def input_path_rows(rank, world, dev, B=32, steps=20, warmup=3):
    """the H2D-inclusive figures beside the headline (never `value`).  Same workload as the headline (FrameAvg,
    bs=32, mixup), a few steps each: (i) int16 waveforms uploaded from page-locked host memory one batch ahead on a copy stream
    (what the train CLI's loader amounts to, reference main.py:238-239 move_data_to_device); (ii) the train CLI's whole input
    pipeline in the loop (seed-1234 sampler -> PinnedBatchLoader threads -> pinned int16 -> copy stream; reference main.py:160-170)."""
    out = []
    for tag, kw in (("H2D-inclusive: Cnn_9layers_FrameAvg B=32 mixup, int16 pinned host -> device every step", {"h2d": True}),
                    ("loader in the loop: Cnn_9layers_FrameAvg B=32 mixup through utils/data_generator.PinnedBatchLoader", {"loader_clips": 192})):
        w = None
        try:
            w = Workload("Cnn_9layers_FrameAvg", B, True, rank, world, dev, **kw)
            dt, loss, _ = w.run(steps, warmup)
            out.append({"config": tag, "workload": w.describe(), "value": round(B * steps / dt, 2), "unit": "clips/s", "steps": steps,
                        "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "metric": "training clips/sec, input path inside the step",
                        "waveform_bytes_per_step_over_pcie": int(w.B2 * 320000 * 2), "loss": round(loss, 5)})
        except Exception as e:                 # a side number must never lose the headline line
            out.append({"config": tag, "value": None, "error": repr(e)})
        finally:
            if w is not None:
                w.close()
            del w
            gc.collect()
            torch.cuda.empty_cache()
    return out


def extra_configs(rank, world, dev, steps=5, warmup=2, hip_graph="off"):
    """The other BASELINE.json configurations, a few steps each (single GPU, same process, after the headline run):
    configs[1] (B=256, the batch north_star's targets are quoted at) first, with its full roofline / front-end / traffic /
    kernel objects."""
    out = []
    try:
        w = Workload("Cnn_9layers_FrameAvg", 256, True, rank, world, dev)
        row, _ = measure(w, 2 * steps, warmup + 1, True, "pmc_traffic.json")
        row = dict({"config": "configs[1] Cnn_9layers_FrameAvg B=256 mixup", "workload": w.describe() + "; BASELINE.json configs[1]",
                    "metric": "training clips/sec"}, **row)
        out.append(row)
        del w
    except Exception as e:
        out.append({"config": "configs[1] Cnn_9layers_FrameAvg B=256 mixup", "value": None, "error": repr(e)})
    gc.collect()
    torch.cuda.empty_cache()
    for tag, mt, B, mix, inf in (
            ("configs[2] Cnn_9layers_FrameAtt B=256 mixup", "Cnn_9layers_FrameAtt", 256, True, False),
            ("configs[3] Cnn_9layers_Gru_FrameAtt B=256 mixup", "Cnn_9layers_Gru_FrameAtt", 256, True, False),
            ("SURVEY 8(f)4 Transformer heads: Cnn_9layers_Transformer_FrameAvg B=256 mixup", "Cnn_9layers_Transformer_FrameAvg", 256, True, False),
            ("configs[0] shape on the GPU: Cnn_9layers_FrameAvg B=32 no mixup", "Cnn_9layers_FrameAvg", 32, False, False),
            ("small per-GPU batch (--batch_size 32 over 8 GPUs in the CLI): Cnn_9layers_FrameAvg B=4 mixup", "Cnn_9layers_FrameAvg", 4, True, False),
            ("inference (eval-mode forward) Cnn_9layers_FrameAvg 256 clips/step", "Cnn_9layers_FrameAvg", 256, False, True)):
        try:
            # the small-per-GPU-batch row always carries BOTH forms (eager kernel-by-kernel and one HIP graph per step) with
            # their un-throttled host enqueue times: that is the regime where the host could become the limiter
            mode = "on" if (B <= 8 and not inf) else hip_graph
            wu = max(warmup, 4) if mode == "on" else warmup
            w = Workload(mt, B, mix, rank, world, dev, inference=inf, hip_graph=graph_eager_steps(mode, B, world, wu, inf))
            k = steps * (8 if B <= 32 else 1)
            dt, loss, _ = w.run(k, wu)
            row = {"config": tag, "workload": w.describe(), "value": round(B * k / dt, 2), "unit": "clips/s", "steps": k,
                   "warmup": warmup, "ms_per_step": round(dt / k * 1e3, 3),
                   "metric": "inference clips/sec" if inf else "training clips/sec", "loss": round(loss, 5)}
            row.update(w.graph_info())
            row["host_enqueue_ms_per_step"] = w.host_enqueue_ms()
            if w.graphed is not None:          # the same steps launched kernel by kernel, for comparison
                w.graphed.enabled = False
                dte, _, _ = w.run(k, 1)
                row["eager_ms_per_step"] = round(dte / k * 1e3, 3)
                row["eager_host_enqueue_ms_per_step"] = w.host_enqueue_ms()
            out.append(row)
            del w
        except Exception as e:                 # a side number must never lose the headline line
            out.append({"config": tag, "value": None, "error": repr(e)})
        gc.collect()
        torch.cuda.empty_cache()
    out.extend(input_path_rows(rank, world, dev))
    return out


def strict_fp32(mt, B, mix, rank, world, dev, steps=5, warmup=2):
    """The same workload with every convolution on the fp32 MFMA pipe (ops.USE_SF16 = False: Winograd F(2x2,3x3)) -- the
    strict-fp32 companion of the headline number."""
    prev, ops.USE_SF16 = ops.USE_SF16, False
    try:
        w = Workload(mt, B, mix, rank, world, dev)
        dt, loss, _ = w.run(steps, warmup)
        return {"value": round(B * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
                "warmup": warmup, "loss": round(loss, 5),
                "arithmetic": "fp32 operands on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), fused Winograd F(2x2,3x3)"}
    except Exception as e:
        return {"value": None, "error": repr(e)}
    finally:
        ops.USE_SF16 = prev
        gc.collect()
        torch.cuda.empty_cache()


class StrongRowWatchdog(object):
    """Started on EVERY rank right before the strong-scaling row of an N > 1 run, cancelled when the row is back.  If it is not back
    after `seconds` (a collective that never completes): rank 0 prints the finished headline line with `strong.error`, and every
    rank ends the process with exit code 0 -- the headline measurement is never lost to a side row."""

    def __init__(self, line, seconds):
        import threading
        self.line, self.seconds = line, seconds
        self.timer = threading.Timer(seconds + (0.0 if line is not None else 3.0), self.fire)     # rank 0 first, the others 3 s later
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        self.timer.cancel()

    def fire(self):
        if self.line is not None:
            if "strong" not in self.line:          # (a row that DID come back is kept: then it is the shutdown that hangs)
                self.line["strong"] = {"scaling": "strong", "value": None,
                                       "error": "watchdog: the strong-scaling row did not finish within %g s" % self.seconds}
            else:
                self.line["dist"]["shutdown_error"] = "watchdog: the process group did not shut down within %g s" % self.seconds
            sys.stdout.write("\n" + json.dumps(self.line) + "\n")
            sys.stdout.flush()
        os._exit(0)


def ranks_seen(dev):
    """Proof that the backend really spans the ranks the line claims: all-reduce (sum) of a 1 from every rank."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return 1
    t = torch.ones((1,), device=dev, dtype=torch.float32)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return int(round(float(t.item())))


def dist_report(wl, steps):
    """`dist` object of a row: backend, ranks the backend saw, exposed all-reduce time of the TIMED steps, bucket bytes."""
    b = wl.opt.buckets
    # wait_events collects one pair per optimiser step from the first warm-up step on: the timed region is the last `steps`
    # entries of the first run() (later passes -- kernel table, un-evented rerun -- append behind them)
    waits = (b.wait_events or [])
    first = getattr(wl, "first_run_steps", None)
    waits = waits[first[0]:first[0] + first[1]] if first else waits[-steps:]
    init = torch.distributed.is_available() and torch.distributed.is_initialized()
    return {"backend": torch.distributed.get_backend() if init else None,
            "world_size": torch.distributed.get_world_size() if init else 1,
            "n_ranks_seen": ranks_seen(wl.dev),
            # time the compute stream sat behind the bucketed all-reduces in optimizer.step() = the EXPOSED part of the
            # gradient exchange (the rest ran beside the backward pass), averaged over the timed steps; null at 1 rank
            "allreduce_exposed_ms_per_step": (round(sum(a.elapsed_time(c) for a, c in waits) / max(len(waits), 1), 4)
                                              if waits else None),
            "allreduce_exposed_steps_averaged": len(waits),
            # SED_ALLREDUCE_OVERLAP=0: every bucket goes out behind the backward pass instead of from inside it
            "allreduce_overlap": not (b.deferred or wl.graphed is not None),
            "nonfinite_poll_lag_steps": wl.opt.poll_lag,
            "flat_gradient_bytes": int(wl.opt.flat_grad.numel() * 4),
            "bucket_bytes": [int(4 * (hi - lo)) for lo, hi in b.ranges]}


def strong_row(mt, B_local, mix, rank, world, dev, steps, warmup, seconds=10, int16=False):
    """Strong-scaling reading of the metric at N ranks: the reference's `--batch_size 32` is the GLOBAL batch that
    nn.DataParallel scatters (pytorch/main.py:138, :160-166), i.e. 32 / N clips per GPU.  Small per-GPU batches are launch-bound,
    so forward + loss + backward replay as one HIP graph per step (graph.GraphedTrainStep; the bucketed all-reduce then goes out
    behind the graph, in optimizer.step())."""
    try:
        wu = max(warmup, 4)
        w = Workload(mt, B_local, mix, rank, world, dev, seconds=seconds, int16=int16, hip_graph=min(3, wu - 1))
        w.opt.buckets.wait_events = []
        dt, loss, _ = w.run(steps, wu)
        row = {"metric": "training clips/sec (10s@32kHz) %s GLOBAL bs=%d at %d GPU (%d clips per GPU)" % (mt, B_local * world, world, B_local),
               "scaling": "strong", "value": round(B_local * world * steps / dt, 2), "unit": "clips/s", "steps": steps, "warmup": wu,
               "ms_per_step": round(dt / steps * 1e3, 3), "per_gpu_batch": B_local, "global_batch": B_local * world,
               "workload": w.describe(seconds, int16), "loss": round(loss, 5), "dist": dist_report(w, steps)}
        row.update(w.graph_info())
        w.opt.buckets.wait_events = None
        del w
        return row
    except Exception as e:                 # a side row must never lose the headline line -- but every rank must get here together
        return {"scaling": "strong", "value": None, "error": repr(e)}
    finally:
        gc.collect()
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model_type", type=str, default="Cnn_9layers_FrameAvg",
                    help="Cnn_9layers_Gru_FrameAtt = BASELINE.json configs[3] / [4] (with --gpus 8 --batch_size 256)")
    ap.add_argument("--batch_size", type=int, default=32,
                    help="post-mixup clips per GPU per step (32 = the configuration BASELINE.json's metric is quoted on; 256 = configs[1])")
    ap.add_argument("--no_mixup", action="store_true")
    ap.add_argument("--seconds", type=int, default=10)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_strong", action="store_true", help="skip the strong-scaling row of a --gpus N > 1 run")
    ap.add_argument("--no_extra", action="store_true", help="skip the extra_configs runs (other BASELINE.json configurations)")
    ap.add_argument("--int16", action="store_true", help="feed int16 waveforms (the HDF5 storage dtype)")
    ap.add_argument("--by_shape", action="store_true", help="print a per-layer MFMA kernel table to stderr")
    ap.add_argument("--no_kernel_events", action="store_true",
                    help="diagnostic: no HIP event pairs inside the timed region (roofline = null)")
    ap.add_argument("--hip_graph", type=str, default="off", choices=("on", "off"),
                    help="replay forward+backward as one HIP graph per step (graph.GraphedTrainStep; one rank only)")
    ap.add_argument("--cpu_threads", type=int, default=0)
    ap.add_argument("--inference", action="store_true",
                    help="secondary metric (SURVEY.md 8d): eval-mode forward only, clips/s over --batch_size waveforms per step")
    ap.add_argument("--h2d", action="store_true",
                    help="secondary: each step first copies its int16 waveforms from pinned host memory (PCIe-inclusive rate)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks (one per GPU) or fail loudly -- never a silent 1-rank run
        parallel.respawn_under_torchrun(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    # SED_SHARE_GPU=1 (tests only, parallel.init_from_env): all ranks share GPU 0 and talk over gloo, so that the N-rank
    # code path of this script can be exercised on a 1-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    share_gpu = os.environ.get("SED_SHARE_GPU") == "1"
    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP kernels only; no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.batch_size
    mix = not args.no_mixup
    ge = graph_eager_steps(args.hip_graph, B, world, args.warmup, args.inference, args.h2d)
    wl = Workload(args.model_type, B, mix, rank, world, dev, seconds=args.seconds, inference=args.inference, int16=args.int16,
                  h2d=args.h2d, hip_graph=ge)
    B2 = wl.B2
    plain = (mix and args.model_type == "Cnn_9layers_FrameAvg" and not args.inference and not args.h2d and not args.int16
             and args.seconds == 10)
    default_workload = plain and B == 32                   # the configuration the metric is quoted on
    pmc_file = "pmc_traffic_b32.json" if B == 32 else "pmc_traffic.json"
    has_pmc = plain and B in (32, 256)
    wl.opt.buckets.wait_events = []   # HIP events around the compute stream's wait for the gradient all-reduces
    frames = 32000 * args.seconds // 320 + 1
    if args.inference or ge:
        # no per-kernel events: graph replays cannot carry them, and the inference metric is a secondary one
        dt, loss, _ = wl.run(args.steps, args.warmup)
        row = {"roofline": None, "roofline_frontend": None, "kernels": {}}
        row.update(wl.graph_info())
        if world == 1:
            row["host_enqueue_ms_per_step"] = wl.host_enqueue_ms()
            if wl.graphed is not None and not args.no_kernel_events:
                wl.graphed.enabled = False
                kern, roof, fe, ms2 = wl.kernel_table(min(args.steps, 10), has_pmc, pmc_file, by_shape=args.by_shape, frames=frames)
                row.update({"roofline": roof, "roofline_frontend": fe, "kernels": kern, "eager_ms_per_step_with_kernel_events": round(ms2, 3)})
    else:
        row, dt = measure(wl, args.steps, args.warmup, has_pmc, pmc_file, by_shape=args.by_shape, frames=frames,
                          events_in_region=not args.no_kernel_events)
        loss = row["loss"]
    bucket_order = list(wl.opt.buckets.last_issue_order)
    dist_info = dist_report(wl, args.steps)
    wl.opt.buckets.wait_events = None
    bucket_ranges = [[lo, hi] for lo, hi in wl.opt.buckets.ranges]
    clips_per_s = B * world * args.steps / dt
    line = {
        # BASELINE.json: "training clips/sec (10s@32kHz) Cnn_9layers_FrameAvg bs=32 at 1/2/4/8 GPU"
        "metric": ("inference clips/sec (10s@32kHz, eval mode) %s bs=%d per GPU at %d GPU" % (args.model_type, B, world))
                  if args.inference else
                  ("training clips/sec (10s@32kHz) %s bs=%d at %d GPU" % (args.model_type, B, world)),
        "value": round(clips_per_s, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (split-f16x3 MFMA products, fp32 accumulate)" if ops.USE_SF16 else "f32", "data": "synthetic",
        "arithmetic": ("fp32 storage and accumulation throughout; 3x3 convolution products (forward, dgrad, weight gradients) as "
                       "three split-f16 MFMAs with fp32 accumulation, operand scales from device-side amax values -- the "
                       "rounding error of a direct fp32 convolution at any magnitude (tests/test_gpu_sf16.py vs float64); "
                       "`strict_fp32` = the same workload on the fp32 MFMA pipe")
                      if ops.USE_SF16 else "fp32 throughout (fp32 MFMA, Winograd F(2x2,3x3))",
        "config": {"workload": wl.describe(args.seconds, args.int16) + ("; the configuration BASELINE.json's metric is quoted on "
                                                                         "(bs=32 per GPU, weak scaling)" if default_workload else
                                                                         ("; BASELINE.json configs[1]" if (plain and B == 256) else " (modified by flags)")),
                   "global_batch": B * world, "waveforms_per_step": B2 * world,
                   "parallelism": "dp%d" % world + (" (TEST MODE: ranks share one GPU over gloo)" if share_gpu else ""),
                   "grad_allreduce": "%d buckets of the flat fp32 gradient (elements %s), issued %s in the "
                                     "order %s" % (len(bucket_ranges), bucket_ranges,
                                                   "behind the backward pass" if wl.opt.buckets.deferred else "from inside backward",
                                                   bucket_order)},
        "waveforms_per_s": round(B2 * world * args.steps / dt, 2),
        "loss": round(float(loss), 5),
        "dist": dist_info,
    }
    for k, v in row.items():
        if k not in ("value", "unit", "steps", "warmup", "ms_per_step", "loss"):
            line[k] = v
    line.setdefault("hip_graph", False)
    # ---- N > 1 only: the strong-scaling row.  The headline line is COMPLETE before it starts, and a watchdog on every rank
    # guarantees that a row that does not come back (a collective that never completes on hardware this code has not seen)
    # costs the row, not the line: rank 0 prints the headline with `strong.error`, every rank leaves with exit code 0
    if world > 1 and B == 32 and mix and not args.inference and not args.h2d and 32 % world == 0 and not args.no_strong:
        # the OTHER reading of the metric: --batch_size 32 is the GLOBAL batch the reference's DataParallel scatters
        # (pytorch/main.py:138, :160-166): 32 / N clips per GPU.  Every rank runs it (collectives), rank 0 reports it.
        line["cpu_baseline"] = None
        guard = StrongRowWatchdog(line if rank == 0 else None, float(os.environ.get("SED_BENCH_STRONG_TIMEOUT_S", "300")))
        line["strong"] = strong_row(args.model_type, 32 // world, mix, rank, world, dev, args.steps, args.warmup, args.seconds, args.int16)
        try:
            parallel.shutdown()       # under the same watchdog: a side row that left the backend in a bad state must not cost the line
        except Exception as e:
            line["dist"]["shutdown_error"] = repr(e)
        guard.cancel()
    else:
        parallel.shutdown()           # all ranks: barrier + destroy the process group; rank 0 then reports alone
    if rank != 0:
        return
    del wl
    gc.collect()
    torch.cuda.empty_cache()
    if world == 1 and default_workload and not args.no_extra:
        if ops.USE_SF16:
            line["strict_fp32"] = strict_fp32(args.model_type, B, mix, rank, world, dev, steps=20, warmup=3)
            line["value_strict_fp32"] = line["strict_fp32"].get("value")    # the figure with no precision question, top level
        line["extra_configs"] = extra_configs(rank, world, dev, hip_graph=args.hip_graph)
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(threads=args.cpu_threads)
        except Exception as e:                     # the baseline is a reported side number: never lose the bench line
            line["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    else:
        line["cpu_baseline"] = None
    # N > 1: the ranks share this stdout with the native collectives library, which writes unbuffered fragments into it
    # (profiles/r05/two_rank_stdout_diag.txt): start on a fresh line so that the JSON line is a line of its own
    sys.stdout.write(("\n" if world > 1 else "") + json.dumps(line) + "\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
