"""Tensor-level wrappers over the C ABI (include/sed_hip.h) + the autograd Functions of the hot path.

PyTorch is used here only for device memory (caching allocator), the current HIP stream and autograd
bookkeeping; every computation is a hand-written HIP kernel in libsed_hip.so.  There is no CPU path: passing
CPU tensors raises.
"""
import ctypes
import os
import math
import weakref

import numpy as np
import torch

from . import _lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

USE_SF16 = os.environ.get("SED_USE_SF16", "1") != "0"   # split-f16 MFMA convolution (forward / dgrad) where supported; 0: fp32 Winograd kernels
USE_WINOGRAD = 2          # fp32-MFMA path (USE_SF16 off): 2 = fused 2-D Winograd F(2x2,3x3) where supported, else the direct implicit GEMM; 0: direct only
# ConvBlock backward: BN2's (sum dy, sum dy*xhat) from the pooled output + per-window ReLU counts instead of a pass over
# the full-resolution conv output.  The identity divides by gamma, so the kernels themselves fall back to the exact pass
# (on the device, from this step's weights) whenever some |gamma| < POOL_BWD_GAMMA_MIN.
POOL_BWD_WINDOWED = True
POOL_BWD_GAMMA_MIN = 1e-2
USE_FUSED_GRU = True        # False: per-step GEMM + gate launches (any hidden size; the fused step kernels are built for 256)

# Weight-gradient kernels (MFMA-bound, off the critical path of backward: nothing downstream reads dW) are issued on a SIDE
# HIP stream beside the HBM-bound BatchNorm / pool backward passes of the main stream (fork/join, never beside another
# MFMA kernel -- two MFMA kernels sharing the CUs measured 2-4 % slower than back to back, an MFMA kernel + an elementwise
# pass 0-8 % faster: tools/stream_overlap_probe.py).  The MFMA kernel owns every VGPR of the CUs it runs on, so the short
# kernels only get the slots its workgroups free as they retire: the step gains 1.0-1.6 % (A/B on one box), not the 5 % a
# perfect overlap would give.  SED_WGRAD_SIDE_STREAM=0 turns it off (A/B runs).
WGRAD_SIDE_STREAM = os.environ.get("SED_WGRAD_SIDE_STREAM", "1") != "0"
EVAL_POOL_FUSION = os.environ.get("SED_EVAL_POOL_FUSION", "1") != "0"     # inference: pool inside the conv2 epilogue (csrc/conv_sf16.hip)
# Block 1 in training (round 4): the raw conv1 output y1 (the largest tensor of the model: 4.2 GB at batch 256) is never
# materialised.  A statistics pass of the Cin = 1 convolution feeds bn1; a second pass writes a1 = relu(bn1(y1)) ONCE, already as
# the split-f16 operand pairs conv2's forward and weight-gradient kernels would otherwise re-derive from y1 in every tile
# (same bytes as y1; their staging becomes a plain copy, the MFMA operands are bit-identical); conv2's dgrad epilogue and
# conv1's backward recompute the raw y1 they need from the one-channel input (same fma sequence: same bits).  0: round-3 dataflow.
# Measured (profiles/r04): conv2 forward -0.20 ms, its weight gradient -0.21 ms, conv1 backward -0.24 ms at batch 256, but the two
# conv1 passes cost +0.5 ms over the single one (both are bound by their x0 gathers, not by the 4.2 GB write) and the
# recomputing dgrad epilogue +0.27 ms: +0.1 ms net -- built, bit-identical, OFF by default.
B1_ACT_PAIRS = os.environ.get("SED_B1_ACT_PAIRS", "0") != "0"
# Gradients as split-f16 operand pairs (round 4): the two BatchNorm-backward apply passes of a ConvBlock write the tensors that only
# the split-f16 dgrad / weight-gradient kernels read; as pairs (same bytes) those kernels' staging is a plain copy.  The scale
# comes from an upper BOUND of the tensor's amax, computed on the device before the pass (sed_grad_bound).  0: fp32 tensors.
GRAD_PAIRS = os.environ.get("SED_GRAD_PAIRS", "1") != "0"
# ... and the pooled outputs of blocks 1-3 (read only by the next block's conv1, forward and weight gradient, and by the
# block's own windowed backward pass): requested by the models' trunk (ConvBlock.forward(pairs_out=True))
ACT_PAIRS = os.environ.get("SED_ACT_PAIRS", "1") != "0"
# data_ptr of an input gradient a ConvBlock returned -> (its amax vector, numel): the bound of the NEXT ConvBlock backward of the
# same backward pass.  Entries never outlive the pass that made them (an end-of-pass callback clears the table): an address is
# not an identity, and a later tensor at the same address must never inherit a dead tensor's amax.
_GRAD_AMAX = {}
_SIDE = {}
_PENDING = []            # [(event recorded on the side stream, sink or None)] of weight gradients not yet joined


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        # default priority: the device offers (0, -1) only, i.e. nothing BELOW the main stream, and a high-priority side
        # stream starves the main one (measured 265 instead of 95 ms/step)
        st = _SIDE[key] = torch.cuda.Stream(device=key)
    return st


_STREAM_OBJ = {}


def _current_stream_obj():
    """torch.cuda.current_stream() without its 11 us of Python: the Stream object is cached per raw handle."""
    raw = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    st = _STREAM_OBJ.get(raw)
    if st is None:
        if len(_STREAM_OBJ) > 64:
            _STREAM_OBJ.clear()
        st = _STREAM_OBJ[raw] = torch.cuda.current_stream()
    return st


def join_side_stream():
    """Main stream waits for every weight gradient issued on the side stream; their sinks report ready (the bucketed
    all-reduce may fire here) and the tensors they used are released.  Called at the join points of ConvBlockFn.backward
    and once when the backward pass ends."""
    if not _PENDING:
        return
    main = _current_stream_obj()
    pend = list(_PENDING)
    del _PENDING[:]
    for ev, sink, keep in pend:
        main.wait_event(ev)
    for ev, sink, keep in pend:
        if sink is not None:
            sink.done()
    del pend                                     # operands / temporaries die here, in main-stream order BEHIND the wait


def pending_sink_indices(opt=None):
    """Parameter indices (of optimiser `opt`, or of any) whose weight gradient is still un-joined on the side stream."""
    return {sink.index for ev, sink, keep in _PENDING if sink is not None and (opt is None or sink.opt is opt)}


def drop_pending_wgrads():
    """Forget side-stream weight gradients of a backward pass that did not finish (it raised between a fork and its
    join): the main stream waits for the side stream, the entries' tensors are released, and NO sink reports ready --
    the gradients of that pass are void.  Called by FusedAdamAmsgrad.zero_grad()."""
    if _PENDING:
        main = _current_stream_obj()
        for ev, sink, keep in _PENDING:
            main.wait_event(ev)
        del _PENDING[:]
    _GRAD_AMAX.clear()          # amax hand-overs of a backward pass that raised (the engine callback clears them otherwise)


_STREAM_OVERRIDE = None     # set while kernels are being enqueued on the side stream (see _fork_wgrad)


def _fork_wgrad(x, gy, B, H, W, Cin, Cout, in_st=None, sink=None, gy_amax=None, x_amax=None, x_presplit=False, gy_presplit=False):
    """_wgrad on the side stream behind everything enqueued on the main stream so far.  With a sink the gradient lands in
    the flat buffer and is joined later (join_side_stream); without one the tensor is returned after an immediate join.

    Memory: torch's current stream stays the MAIN stream, so every tensor (operands, partial-sum buffers) belongs to the
    main stream's allocator pool; only the kernel launches go to the side stream (_STREAM_OVERRIDE).  All of them are kept
    referenced by the pending entry until the join, after which main-stream-ordered reuse is safe.  (The first version
    used torch.cuda.stream(side) + Tensor.record_stream: correct, but the allocator then parks every such block until
    the side stream's events are polled and reserved memory crept from 78 to 223 GB over 200 steps.)"""
    global _STREAM_OVERRIDE
    main = _current_stream_obj()
    side = _side_stream(x.device)
    if _wgrad_algo(H, W, Cin, Cout) == 3:
        # operand amaxes the caller did not bring are taken HERE, on the main stream: their zeroed pool rows (a fill on
        # torch's current stream) must not be handed to a kernel of the side stream that is not ordered behind the fill
        if gy_amax is None:
            gy_amax = amax_of(gy)
        if x_amax is None:
            x_amax = act_amax_full(x, in_st) if in_st is not None else amax_of(x)
    side.wait_stream(main)
    keep = [x, gy, in_st, gy_amax, x_amax]
    _STREAM_OVERRIDE = side
    try:
        dw = _wgrad(x, gy, B, H, W, Cin, Cout, in_st=in_st, sink=sink, signal=False, gy_amax=gy_amax, x_amax=x_amax, keep=keep,
                    x_presplit=x_presplit, gy_presplit=gy_presplit)
    finally:
        _STREAM_OVERRIDE = None
    ev = torch.cuda.Event()
    ev.record(side)
    if not _PENDING:
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)    # end of this backward pass
    _PENDING.append((ev, sink, keep))
    if sink is None:
        join_side_stream()
        return dw
    return None


# bench.py sets this to a dict to HIP-event-time the MFMA kernels inside its timed region:
# {tag: [(start_event, end_event, algorithmic_flops), ...]}.  None = no instrumentation.  TIMING_ONLY: tuple of tag prefixes
# that are timed (None = every tagged launch) -- an event pair costs the stream ~6 us, so the headline region only brackets
# the kernel family its `roofline` object reports.
TIMING = None
TIMING_ONLY = None


class _timed(object):
    def __init__(self, fmt, args, flops):
        # the tag is only formatted while bench.py is timing (21 conv launches per step would pay for the string otherwise)
        self.on = TIMING is not None
        if self.on:
            self.tag, self.flops = (fmt % args if args is not None else fmt), flops
            self.on = TIMING_ONLY is None or self.tag.startswith(TIMING_ONLY)

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(_STREAM_OVERRIDE)      # the stream the kernel is launched on (None = torch's current stream)

    def __exit__(self, *exc):
        if self.on:
            self.b.record(_STREAM_OVERRIDE)
            TIMING.setdefault(self.tag, []).append((self.a, self.b, self.flops))
        return False


def _ptr(t):
    """Device address for a `void*` / `float*` parameter of the C ABI (plain int: the prototypes installed by _lib convert it)."""
    if t is None:
        return None
    return t.data_ptr()


# Small per-step host arrays (mixup lambdas, SpecAugment stripe tables) go up through a ring of PINNED staging buffers with a
# truly asynchronous copy.  `tensor.to(device)` from pageable memory blocks the host until the copy has run, i.e. until the
# GPU has drained everything enqueued before it: once per step the CPU lost its lead and the GPU idled while the first
# kernels of the next step were being launched (~0.3 ms of a 9 ms step at batch 32).
_PIN_RING = {}


def upload_small(arr, device, dtype=torch.float32, slots=8):
    a = torch.as_tensor(np.ascontiguousarray(arr)).to(dtype).contiguous()            # host tensor (cheap: <= a few KB)
    dev = torch.device(device)
    key = (dev.index, dtype, a.numel())
    ring = _PIN_RING.get(key)
    if ring is None:
        ring = _PIN_RING[key] = {"i": 0, "buf": [torch.empty((a.numel(),), dtype=dtype).pin_memory() for _ in range(slots)],
                                 "ev": [None] * slots}
    k = ring["i"] % slots
    ring["i"] += 1
    if ring["ev"][k] is not None:
        ring["ev"][k].synchronize()                  # the copy that last used this slot (8 uploads ago) is long done
    ring["buf"][k].copy_(a.view(-1))
    out = torch.empty(a.shape, dtype=dtype, device=dev)
    out.view(-1).copy_(ring["buf"][k], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ring["ev"][k] = ev
    return out


def _stream():
    """hipStream_t the next kernel goes to: the side stream while _STREAM_OVERRIDE is set, else torch's current stream of the
    current device.  (torch.cuda.current_stream() builds a Stream object through four Python layers: 11 us per call, a third of
    the host time of a training step at ~100 launches; the raw query is 0.3 us -- tools/host_profile.py.)"""
    if _STREAM_OVERRIDE is not None:
        return _STREAM_OVERRIDE.cuda_stream
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sound_event_detection_dcase2017_task4_amd: the hot path runs on the GPU only "
                               "(HIP kernels, gfx950); got a CPU tensor.  Move model and inputs to 'cuda'.")


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_FN = {}


def _call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.lib(), name)
    rc = fn(*args)
    if rc:
        _lib.check(rc, name)


# ------------------------------------------------------------------------------------------------------------
# asynchronous device-side error flag (fused GRU give-up).  The flag lives in host-mapped pinned memory: kernels
# raise it with a system-scope store, the host polls it without synchronising.

_ERR_FLAG = None
_ERR_DEV = {}
_GUARDED = weakref.WeakSet()      # optimisers whose Adam kernel counts refused steps in their own device word (`_skipped`)


class NonFiniteOperand(RuntimeError):
    """A split-f16 convolution met a NaN / inf operand.  The Adam kernel refuses every step from that point on (parameters
    and moments stay as they were before the poisoned step) until the error has been reported through
    check_device_errors(); `skipped_steps` says how many optimiser steps that were."""

    def __init__(self, msg, skipped_steps=0):
        super(NonFiniteOperand, self).__init__(msg)
        self.skipped_steps = skipped_steps


def _err_flag():
    """One pinned int32[4] per process; its device-visible address is handed to the kernels that can fail at run time.
    Slot 0: fused GRU give-up, slot 1: non-finite operand of a split-f16 convolution."""
    global _ERR_FLAG
    if _ERR_FLAG is None:
        _ERR_FLAG = torch.zeros((4,), dtype=torch.int32).pin_memory()
    return _ERR_FLAG


def _err_dev(device=None):
    """Device int32[2] per GPU: [0] = non-finite operand seen (written by the split-f16 kernels, read by the Adam kernel as
    its skip flag), [1] = optimiser steps the Adam kernel refused since."""
    idx = torch.device(device).index if device is not None and torch.device(device).index is not None \
        else torch.cuda.current_device()
    t = _ERR_DEV.get(idx)
    if t is None:
        t = _ERR_DEV[idx] = torch.zeros((2,), dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def _sf16_err_ptr():
    """Host-mapped word the split-f16 convolutions set when an operand is not finite (slot 1 of the flag)."""
    return ctypes.c_void_p(_err_flag().data_ptr() + 4)


def _sf16_err_dev_ptr(device=None):
    return ctypes.c_void_p(_err_dev(device).data_ptr())


def clear_nonfinite_flags():
    """Reset the found-non-finite words (host-mapped + device) after they have been reported."""
    for t in _ERR_DEV.values():
        t.zero_()
    for t in _TICKETS.values():        # split-K tickets are self-resetting -- unless a launch was aborted half-way: after an error
        t.zero_()                      # report they are zero again, so no later launch can be left without its last arriver
    torch.cuda.synchronize()
    if _ERR_FLAG is not None:
        _ERR_FLAG[1] = 0


def check_device_errors(synchronize=False, nonfinite=True):
    """Raise if a kernel reported a run-time failure since the last check (no device synchronisation unless asked: the
    flag is written through host-mapped memory, so a failure surfaces at the next call after the kernel ran).  Only the
    slot that is reported is cleared.

    nonfinite: whether the found-non-finite word is reported HERE.  Callers that only care about the other flags pass False
    (the fused GRU's launch check; an optimiser that polls the word itself at a deterministic point of its step,
    FusedAdamAmsgrad(poll_lag=...): every rank of a data-parallel job must learn about a refused step at the same
    iteration, so an opportunistic poll must not pre-empt it)."""
    if _ERR_FLAG is None:
        return
    if synchronize:
        torch.cuda.synchronize()
    if nonfinite and int(_ERR_FLAG[1]):
        torch.cuda.synchronize()                      # rare path: settle, then read how many steps the Adam kernel refused
        skipped = 0
        for opt in list(_GUARDED):                    # every guarded optimiser takes ITS refused steps back, whoever polls
            k = int(opt._skipped.item())
            if k:
                opt._skipped.zero_()
                opt.step_count = max(0, opt.step_count - k)
                opt.skipped_steps += k
                skipped = max(skipped, k)
        for t in _ERR_DEV.values():
            skipped = max(skipped, int(t[1].item()))  # raw users of adam_amsgrad_ (no optimiser object)
        clear_nonfinite_flags()
        raise NonFiniteOperand(
            "sound_event_detection_dcase2017_task4_amd: a split-f16 convolution met a NaN / inf operand (diverged training or "
            "non-finite input): its results are NaN.  The Adam kernel refused the %d optimiser step(s) since -- parameters and "
            "moments are those from before the poisoned step.  (Operand scales come from device-side amax values, so a FINITE "
            "operand cannot overflow at any magnitude.)  BatchNorm running statistics were not touched by the poisoned "
            "forward passes either; roll `num_batches_tracked` back with ops.rollback_bn_counters(model, n).  The train CLI "
            "re-runs such steps on the fp32 MFMA kernels (ops.USE_SF16 = False), which propagate NaN exactly like the "
            "reference." % skipped, skipped)


    code = int(_ERR_FLAG[0])
    if code:
        _ERR_FLAG[0] = 0
        raise RuntimeError(
            "sound_event_detection_dcase2017_task4_amd: the fused GRU recurrence (sed_gru_seq_%s) gave up waiting for its "
            "partner workgroups -- its 128 persistent workgroups were not all resident (CU mask, partitioned GPU or a "
            "co-tenant kernel).  Its outputs were overwritten with NaN.  Set ops.USE_FUSED_GRU = False to use the "
            "per-step launches on this device." % ("fwd" if code == 1 else "bwd"))


def rollback_bn_counters(model, n):
    """Take `n` refused training steps back out of every BatchNorm `num_batches_tracked` of `model` (the forward pass of a
    refused step bumped them; the running statistics themselves were guarded on the device)."""
    bufs = model.bn_counters() if hasattr(model, "bn_counters") else [
        b for name, b in model.named_buffers() if name.endswith("num_batches_tracked")]
    if n and bufs:
        torch._foreach_sub_(bufs, int(n))


# ------------------------------------------------------------------------------------------------------------
# gradient sinks (optim.FusedAdamAmsgrad, direct_grads): the backward kernels write weight gradients straight into
# the flat gradient buffer and autograd gets None for them -- no per-parameter accumulate kernels.

def _sinks(ctx, params, first_index):
    """Remember, for the parameters passed to a Function.forward at positions first_index.., where their gradients go
    (None = the ordinary autograd return path) and tell the bucketed all-reduce to expect them."""
    out = []
    for k, p in enumerate(params):
        s = getattr(p, "_sed_sink", None) if p is not None else None
        if s is not None and not ctx.needs_input_grad[first_index + k]:
            s = None
        if s is not None:
            s.expect()
        out.append(s)
    return out


def _dst(sink, shape, device):
    """Tensor a gradient kernel should write into."""
    if sink is not None:
        return sink.view.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _ret(sink, t):
    """What backward() returns for a gradient already written into _dst(sink, ...)."""
    if sink is not None:
        sink.done()
        return None
    return t


def _put(sink, t):
    """Return path for a gradient that was computed into a temporary (slices of padded / stacked results)."""
    if sink is not None:
        sink.view.copy_(t.reshape(sink.view.shape))
        sink.done()
        return None
    return t.contiguous()


# weight operands derived from parameters (padded head matrices, stacked GRU input weights): rebuilt only when a
# source parameter changed.  In-place updates through torch bump `_version`; the fused Adam kernel updates the flat
# buffer through raw pointers and bumps PARAM_GENERATION instead (so does invalidate_weight_caches()).
PARAM_GENERATION = 0
_WCACHE = {}


def invalidate_weight_caches():
    global PARAM_GENERATION
    PARAM_GENERATION += 1


def _cache_stamp(srcs):
    return (PARAM_GENERATION,) + tuple((t.data_ptr(), t._version) for t in srcs)


def _cache_fresh(kind, srcs):
    hit = _WCACHE.get((kind,) + tuple(id(t) for t in srcs))
    return hit is not None and hit[0] == _cache_stamp(srcs) and all(r() is t for r, t in zip(hit[2], srcs))


def _cache_trim():
    """Keep the table small WITHOUT touching live entries of the current parameter generation: entries whose tensors are gone or
    that belong to an older generation go first; only a table that is still large after that is dropped.  (Clearing everything at a
    fixed size could wipe entries that the very same prepack_sf16 call had just put -- found as a test that failed once in five
    runs, whenever earlier tests had left the table near the limit.)"""
    if len(_WCACHE) <= 64:
        return
    for k in [k for k, (stamp, _, refs) in _WCACHE.items() if stamp[0] != PARAM_GENERATION or any(r() is None for r in refs)]:
        del _WCACHE[k]
    if len(_WCACHE) > 512:
        _WCACHE.clear()


def _cache_put(kind, srcs, val):
    _cache_trim()
    _WCACHE[(kind,) + tuple(id(t) for t in srcs)] = (_cache_stamp(srcs), val, [weakref.ref(t) for t in srcs])


def _cached(kind, srcs, build):
    """The entry belongs to these very tensor OBJECTS (weak references: the allocator recycles addresses and Python
    recycles ids, so neither identifies a parameter) in this very state (storage address, torch version counter,
    PARAM_GENERATION)."""
    key = (kind,) + tuple(id(t) for t in srcs)
    stamp = (PARAM_GENERATION,) + tuple((t.data_ptr(), t._version) for t in srcs)
    hit = _WCACHE.get(key)
    if hit is not None and hit[0] == stamp and all(r() is t for r, t in zip(hit[2], srcs)):
        return hit[1]
    val = build()
    _cache_trim()
    _WCACHE[key] = (stamp, val, [weakref.ref(t) for t in srcs])
    return val


# ------------------------------------------------------------------------------------------------------------
# front-end constants (host side, built once per model; numpy float64 -> float32)

MEL_TASK_TAPS = 12


def frontend_tables(window, melW, device):
    """window: (1024,) tensor = conv_real.weight[0,0,:]; melW (513,64) tensor.  Returns the device tables the
    log-mel kernel needs (FFT twiddles + the mel filter bank cut into <= 12-tap tasks)."""
    tw = np.exp(-2j * np.pi * (np.arange(32)[None, :] * np.arange(32)[:, None]) / 1024.0)   # [32 k1][32 n2]
    tw1024t = np.stack([tw.real, tw.imag], axis=-1).astype(np.float32)
    W = melW.detach().cpu().numpy().astype(np.float32)                           # (513, 64)
    tasks, bands, vals = mel_task_tables(W)
    dev = torch.device(device)
    return {
        "window": window.detach().to(dev, torch.float32).contiguous(),
        "tw1024t": torch.from_numpy(tw1024t).to(dev).contiguous(),
        "mel_tasks": torch.from_numpy(tasks).to(dev).contiguous(),
        "n_tasks": int(tasks.shape[0]),
        "mel_bands": torch.from_numpy(bands).to(dev).contiguous(),
        "max_band_tasks": int((bands >= 0).sum(axis=1).max()),
        "mel_w": torch.from_numpy(vals).to(dev).contiguous(),
        "mel_nnz": int(vals.size),
    }


def mel_task_tables(W, slots=128):
    """melW (513, 64) -> the mel-stage tables of the log-mel kernel (include/sed_hip.h):

    tasks (slots, 4) int32 {window start bin, taps, offset into weights, band}, bands (64, 4) int32 task slots per band (-1 = none),
    weights (slots * 12,) float32.  The non-zero run of every band is cut into <= 12-bin chunks; each chunk becomes a task that
    reads a 12-bin window CONTAINING it (zero weights outside the chunk).  Where the window starts is free within 12 - len
    positions, and that freedom is used to make the reads of the kernel bank-conflict free: task slot s is handled by lane
    s % 64 in round s // 64, a ds_read_b64 is conflict-free when the 32 lanes of a half-wave read (Pa, Pb) pairs at bins that
    differ mod 32, so the tasks are matched (Kuhn's augmenting paths) to cells (group s // 32, start mod 32), one task per cell."""
    nbins, nbands = W.shape
    if nbands != 64:
        raise RuntimeError("the log-mel kernel is built for 64 mel bands")
    chunks = []                                                   # (band, first bin, last bin + 1)
    for m in range(nbands):
        nz = np.nonzero(W[:, m])[0]
        if len(nz):
            a, b = int(nz[0]), int(nz[-1]) + 1
            for c in range(a, b, MEL_TASK_TAPS):
                chunks.append((m, c, min(c + MEL_TASK_TAPS, b)))
    groups = slots // 32
    if len(chunks) > slots or max(sum(1 for ch in chunks if ch[0] == m) for m in range(nbands)) > 4:
        raise RuntimeError("mel filter bank does not fit the kernel tables (%d tasks)" % len(chunks))
    allowed = [list(range(max(0, e - MEL_TASK_TAPS), c + 1)) for (_, c, e) in chunks]      # window starts that cover the chunk
    cell_of, task_of = {}, {}                                      # task -> (group, residue, start);  (group, residue) -> task

    def try_place(t, seen):
        for st in allowed[t]:
            for g in range(groups):
                cell = (g, st % 32)
                if cell in seen:
                    continue
                seen.add(cell)
                if cell not in task_of or try_place(task_of[cell], seen):
                    task_of[cell] = t
                    cell_of[t] = (g, st % 32, st)
                    return True
        return False

    order = sorted(range(len(chunks)), key=lambda t: len(allowed[t]))          # most constrained first
    placed_all = all(try_place(t, set()) for t in order)
    tasks = np.zeros((slots, 4), dtype=np.int32)
    bands = -np.ones((nbands, 4), dtype=np.int32)
    vals = np.zeros((slots * MEL_TASK_TAPS,), dtype=np.float32)
    next_lane = [0] * groups
    for t, (m, c, e) in enumerate(chunks):
        if placed_all:
            g, _, st = cell_of[t]
        else:                                                      # (never for the reference's filter bank) fall back: any slot
            g, st = min(range(groups), key=lambda q: next_lane[q]), allowed[t][-1]
        slot = g * 32 + next_lane[g]
        next_lane[g] += 1
        tasks[slot] = [st, MEL_TASK_TAPS, slot * MEL_TASK_TAPS, m]
        vals[slot * MEL_TASK_TAPS + (c - st):slot * MEL_TASK_TAPS + (e - st)] = W[c:e, m]
        bands[m, int((bands[m] >= 0).sum())] = slot
    for g in range(groups):                                        # idle slots read too (zero weights): keep them off the used banks
        used = {int(tasks[s][0]) % 32 for s in range(g * 32, g * 32 + next_lane[g])}
        free = [r for r in range(32) if r not in used]
        for i, slot in enumerate(range(g * 32 + next_lane[g], g * 32 + 32)):
            tasks[slot] = [free[i % len(free)] if free else 0, 0, slot * MEL_TASK_TAPS, 0]
    return tasks, bands, vals


def logmel(wave, tables, amin=1e-10):
    """wave (B2, L) float32 or int16 on the GPU -> (B2, T, 64) float32.  models.py:284-285."""
    _chk_dev(wave)
    if wave.dim() != 2:
        raise RuntimeError("waveform must be (batch, samples)")
    wave = wave.contiguous()
    B2, L = wave.shape
    T = L // 320 + 1
    out = torch.empty((B2, T, 64), dtype=torch.float32, device=wave.device)
    name = "sed_logmel_i16" if wave.dtype == torch.int16 else "sed_logmel_f32"
    if wave.dtype not in (torch.int16, torch.float32):
        wave = wave.float()
    in_bytes = 2 if wave.dtype == torch.int16 else 4
    with _timed("logmel_frontend", None, float(B2) * (L * in_bytes + T * 64 * 4)):      # "flops" slot carries ALGORITHMIC BYTES
        _call(name, _ptr(wave), B2, L, _ptr(tables["window"]), _ptr(tables["tw1024t"]),
              _ptr(tables["mel_tasks"]), tables["n_tasks"], _ptr(tables["mel_bands"]), tables["max_band_tasks"],
              _ptr(tables["mel_w"]), tables["mel_nnz"], amin, _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------------------
# BatchNorm helpers

class BnStats(object):
    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, C, device):
        buf = torch.empty((4, C), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def _ws(C, device):
    return torch.empty((2048 * C,), dtype=torch.float64, device=device)


def bn_finalize(partials, nparts, rows_per_part, N, bn_w, bn_b, running_mean, running_var, y_amax=None, act_bound_out=None,
                minmax=None, act_amax_out=None):
    """Batch statistics -> folded affine + running-statistics update.  minmax / act_amax_out: the same launch also leaves the
    amax of relu(scale*y + shift) from the per-part range of y (what act_amax() computes; a follow-up launch beyond 512 parts).  While the found-non-finite guard is on (USE_SF16)
    NaN / inf batch statistics raise the error words (the Adam kernel then refuses the step) and are NOT blended into
    running_mean / running_var: a refused step leaves the BatchNorm buffers intact like the parameters.  With
    USE_SF16 = False the update is torch's (NaN flows into the buffers, as in the reference)."""
    C = bn_w.numel()
    st = BnStats(C, bn_w.device)
    guard = USE_SF16
    cand = torch.empty((2, C), dtype=torch.float32, device=bn_w.device) if (guard and running_mean is not None) else None
    _call("sed_bn_finalize", _ptr(partials), nparts, rows_per_part, N, C, _ptr(bn_w), _ptr(bn_b), BN_EPS, BN_MOMENTUM,
          _ptr(running_mean), _ptr(running_var), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
          _ptr(_ws(C, bn_w.device)), _sf16_err_dev_ptr(bn_w.device) if guard else None, _sf16_err_ptr() if guard else None,
          _ptr(cand), _ptr(y_amax) if act_bound_out is not None else None, _ptr(act_bound_out),
          _ptr(minmax) if act_amax_out is not None else None, _ptr(act_amax_out), _stream())
    if cand is not None:
        # the proposed running statistics are installed by ONE launch at the end of the forward pass (models' trunk:
        # begin_bn_commit / commit_bn) -- or right here for a BatchNorm used on its own -- unless the pass met NaN / inf
        if _BN_COMMIT is not None:
            _BN_COMMIT.append((cand, running_mean, running_var, C))
        else:                                   # a BatchNorm used on its own: installed right away, tracked like the others
            _commit_bn_entries([(cand, running_mean, running_var, C)])
            _track_bn([(cand, running_mean, running_var, C)])
    return st


_BN_COMMIT = None        # [(cand, running_mean, running_var, C)] while a forward pass collects its BatchNorm updates
# Entry groups installed since the last optimiser step, oldest first (one group per forward pass / stand-alone BatchNorm); their
# `cand` now hold the statistics they replaced.  SEVERAL groups exist with gradient accumulation (direct_grads=False) or any second
# train-mode forward before optimizer.step(): a refused step takes ALL of them back, newest first, so that the buffers end up as
# they were before the first pass of the cycle.  Train-mode forward passes that are never followed by an optimiser step (statistics
# recalibration loops) would let the list grow: only the newest _BN_TRACK_MAX groups are kept.
_BN_LAST = None
_BN_TRACK_MAX = 64


def _track_bn(entries):
    global _BN_LAST
    _BN_LAST = ((_BN_LAST or []) + [entries])[-_BN_TRACK_MAX:]


def begin_bn_commit():
    global _BN_COMMIT
    _BN_COMMIT = []


def commit_bn(drop=False):
    """End of a forward pass: install the running statistics its BatchNorms proposed (one launch; skipped on the device when
    the found-non-finite word is set) and keep the ones they replace until the optimiser step has settled
    (restore_bn_if_refused).  drop=True: the pass raised -- forget them."""
    global _BN_COMMIT, _BN_LAST
    entries, _BN_COMMIT = _BN_COMMIT, None
    if entries and not drop:
        for i in range(0, len(entries), 16):
            _commit_bn_entries(entries[i:i + 16])
        _track_bn(entries)


def restore_bn_if_refused():
    """Behind the Adam launch of optim.FusedAdamAmsgrad.step(): when that step was refused for a reason that arose AFTER the
    forward pass (a split-f16 dgrad / wgrad met NaN, the all-reduced gradient is non-finite, another rank raised its flag) the
    running statistics the forward pass installed are taken back on the device -- the found-non-finite word decides, no
    synchronisation.  A refused step leaves the BatchNorm buffers as intact as parameters and moments, on every rank."""
    global _BN_LAST
    groups, _BN_LAST = _BN_LAST, None
    if groups and USE_SF16:
        for entries in reversed(groups):         # newest pass first: the oldest `cand` of a buffer is what it ends up with
            for i in range(0, len(entries), 16):
                _commit_bn_entries(entries[i:i + 16], restore=True)


def _commit_bn_entries(entries, restore=False):
    n = len(entries)
    _call("sed_bn_restore" if restore else "sed_bn_commit", n, (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in entries]),
          (ctypes.c_void_p * n)(*[e[1].data_ptr() for e in entries]), (ctypes.c_void_p * n)(*[e[2].data_ptr() for e in entries]),
          (ctypes.c_int * n)(*[e[3] for e in entries]), _sf16_err_dev_ptr(entries[0][1].device), _stream())


def bn_eval_affine(bn_w, bn_b, running_mean, running_var):
    C = bn_w.numel()
    st = BnStats(C, bn_w.device)
    _call("sed_bn_eval_affine", C, _ptr(bn_w), _ptr(bn_b), _ptr(running_mean), _ptr(running_var), BN_EPS, _ptr(st.mean),
          _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift), _stream())
    return st


def bn_bwd_finalize(partials, nparts, N, st, want_coef=True, batch_stats=True, sinks=(None, None), bound=None, minmax=None):
    """bound (optional) = (y_amax, g_amax, ginv, bound_out): the finalize launch also leaves, in the zeroed amax vector bound_out,
    an upper bound of max |a*dy + b*y + c| for |y| <= amax(y_amax), |dy| <= amax(g_amax) * ginv (what sed_grad_bound computes).
    minmax (then y_amax = None): the per-part range of y [nparts][2][C] -- the bound uses each channel's own range."""
    C = st.mean.numel()
    dev = st.mean.device
    dgamma = _dst(sinks[0], (C,), dev)
    dbeta = _dst(sinks[1], (C,), dev)
    coef = torch.empty((3, C), dtype=torch.float32, device=dev) if want_coef else None
    _call("sed_bn_bwd_finalize", _ptr(partials), nparts, N, C, _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale),
          1 if batch_stats else 0, _ptr(dgamma), _ptr(dbeta), _ptr(coef), _ptr(_ws(C, dev)),
          _ptr(bound[0]) if bound else None, _ptr(bound[1]) if bound else None, float(bound[2]) if bound else 1.0,
          _ptr(bound[3]) if bound else None, _ptr(minmax) if bound else None, _stream())
    return _ret(sinks[0], dgamma), _ret(sinks[1], dbeta), coef


# ------------------------------------------------------------------------------------------------------------
# bn0 + SpecAugment + mixup

class Bn0AugMix(torch.autograd.Function):
    """models.py:287-296.  logmel (B2,T,64) -> x0 (B,T,64) [NHWC with C=1].  Gradients: bn0.weight/bias only."""

    @staticmethod
    def forward(ctx, lm, bn_w, bn_b, running_mean, running_var, training, stripes, lam):
        _chk_dev(lm, bn_w)
        lm = _f32c(lm)
        B2, T, M = lm.shape
        assert M == 64
        dev = lm.device
        if training:
            N = B2 * T
            rpp = _lib.lib().sed_stats_rows_per_part()
            nparts = (N + rpp - 1) // rpp
            partials = torch.empty((nparts, 2, 64), dtype=torch.float32, device=dev)
            _call("sed_chan_stats", _ptr(lm), N, 64, _ptr(partials), _stream())
            st = bn_finalize(partials, nparts, rpp, N, bn_w, bn_b, running_mean, running_var)
        else:
            st = bn_eval_affine(bn_w, bn_b, running_mean, running_var)
            stripes, lam = None, None
        # the C ABI takes plain pointers: sizes are checked here (a short lambda / stripe table would be read out of bounds)
        if lam is not None and (lam.numel() != B2 or B2 % 2):
            raise ValueError("mixup_lambda must hold one weight per waveform of an EVEN batch (pytorch_utils.py:80-93): "
                             "got %d for a batch of %d" % (lam.numel(), B2))
        if stripes is not None and (tuple(stripes.shape) != (B2, 8) or stripes.dtype != torch.int32):
            raise ValueError("specaug_stripes must be int32 (batch, 8) = [tb0, td0, tb1, td1, fb0, fd0, fb1, fd1] per "
                             "waveform: got %s %s for a batch of %d" % (stripes.dtype, tuple(stripes.shape), B2))
        Bout = B2 // 2 if lam is not None else B2
        out = torch.empty((Bout, T, 64), dtype=torch.float32, device=dev)
        _call("sed_bn0_aug_mix_fwd", _ptr(lm), B2, T, _ptr(st.scale), _ptr(st.shift), _ptr(stripes), _ptr(lam), _ptr(out),
              _stream())
        ctx.st, ctx.training = st, training
        ctx.sinks = _sinks(ctx, (bn_w, bn_b), 1)
        ctx.save_for_backward(lm, stripes, lam)
        return out

    @staticmethod
    def backward(ctx, g):
        lm, stripes, lam = ctx.saved_tensors
        st = ctx.st
        g = _f32c(g)
        B2, T, _ = lm.shape
        nparts_max = (B2 * T + 255) // 256
        partials = torch.empty((nparts_max, 2, 64), dtype=torch.float32, device=lm.device)
        n = ctypes.c_int(0)
        _call("sed_bn0_aug_mix_bwd", _ptr(lm), _ptr(g), B2, T, _ptr(st.mean), _ptr(st.invstd), _ptr(stripes), _ptr(lam),
              _ptr(partials), ctypes.byref(n), _stream())
        dgamma, dbeta, _ = bn_bwd_finalize(partials, n.value, B2 * T, st, want_coef=False,   # same formula in eval mode
                                           sinks=ctx.sinks)
        return None, dgamma, dbeta, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------------
# ConvBlock

def _conv_igemm(x, w_packed, B, H, W, Cin, Cout, in_st=None, epi=0, partials=None, yprev=None, p_st=None):
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    with _timed("conv3x3_igemm_mfma(fwd+dgrad)|%d->%d@%dx%d epi%d%s", (Cin, Cout, H, W, epi, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        _call("sed_conv3x3_igemm", _ptr(x), _ptr(w_packed), _ptr(y), B, H, W, Cin, Cout,
              _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None, epi,
              _ptr(partials), _ptr(yprev), _ptr(p_st.scale) if p_st is not None else None,
              _ptr(p_st.shift) if p_st is not None else None, _ptr(p_st.mean) if p_st is not None else None,
              _ptr(p_st.invstd) if p_st is not None else None, _stream())
    return y


def _pack_wino2(w, want_f=True, want_d=False):
    """OIHW -> 2-D Winograd packs uf [Cin/8][16][Cout][8] / ud [Cout/8][16][Cin][8]."""
    Cout, Cin = w.shape[0], w.shape[1]
    uf = torch.empty((Cin // 8, 16, Cout, 8), dtype=torch.float32, device=w.device) if want_f else None
    ud = torch.empty((Cout // 8, 16, Cin, 8), dtype=torch.float32, device=w.device) if want_d else None
    _call("sed_pack_conv_weights_wino2", _ptr(w), Cout, Cin, _ptr(uf), _ptr(ud), _stream())
    return uf, ud


def _wino2_partials(B, H, W, C, device):
    """(buffer, nparts) for the statistics epilogues of sed_conv3x3_wino2: [P][2][C] floats + P counts."""
    P = int(_lib.lib().sed_conv_wino2_num_parts(B, H, W))
    return torch.empty((P * 2 * C + P,), dtype=torch.float32, device=device), P


def _conv_wino2(x, w_wino2, B, H, W, Cin, Cout, in_st=None, epi=0, partials=None, yprev=None, p_st=None):
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    with _timed("conv3x3_wino2d_mfma(fwd+dgrad)|%d->%d@%dx%d epi%d%s", (Cin, Cout, H, W, epi, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        _call("sed_conv3x3_wino2", _ptr(x), _ptr(w_wino2), _ptr(y), B, H, W, Cin, Cout,
              _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None, epi,
              _ptr(partials), _ptr(yprev), _ptr(p_st.scale) if p_st is not None else None,
              _ptr(p_st.shift) if p_st is not None else None, _ptr(p_st.mean) if p_st is not None else None,
              _ptr(p_st.invstd) if p_st is not None else None, _stream())
    return y


def _pack(w, want_f=True, want_d=False):
    Cout, Cin = w.shape[0], w.shape[1]
    wf = torch.empty((9, Cout, Cin), dtype=torch.float32, device=w.device) if want_f else None
    wd = torch.empty((9, Cin, Cout), dtype=torch.float32, device=w.device) if want_d else None
    _call("sed_pack_conv_weights", _ptr(w), Cout, Cin, _ptr(wf), _ptr(wd), _stream())
    return wf, wd


def _wgrad_wino2(x, gy, B, H, W, Cin, Cout, in_st=None, sink=None, signal=True, keep=None):
    ns, ups = ctypes.c_int(0), ctypes.c_int(0)
    nfl = _lib.lib().sed_wgrad_wino2_partial_floats(B, H, W, Cin, Cout, ctypes.byref(ns), ctypes.byref(ups))
    if nfl <= 0:
        raise RuntimeError("sed_conv3x3_wgrad_wino2 does not support this shape")
    partial = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    dw = _dst(sink, (Cout, Cin, 3, 3), x.device)
    if keep is not None:
        keep.extend((partial, dw))                   # alive until the side stream has been joined
    with _timed("conv3x3_wgrad_wino2d_mfma(+slice reduce)|%d->%d@%dx%d%s", (Cin, Cout, H, W, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        _call("sed_conv3x3_wgrad_wino2", _ptr(x), _ptr(gy), _ptr(dw), _ptr(partial), B, H, W, Cin, Cout,
              _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None, _stream())
    return _ret(sink, dw) if signal else (None if sink is not None else dw)


def _wgrad_sf16(x, gy, B, H, W, Cin, Cout, in_st=None, sink=None, signal=True, gy_amax=None, x_amax=None, keep=None,
                x_presplit=False, gy_presplit=False):
    """x_presplit: x holds split-f16 operand pairs (conv1_act_sf16) scaled by x_amax -- no input transform, plain-copy staging."""
    nfl = _lib.lib().sed_wgrad_sf16_partial_floats(B, H, W, Cin, Cout)
    if nfl <= 0:
        raise RuntimeError("sed_conv3x3_wgrad_sf16 does not support this shape")
    if gy_amax is None:
        gy_amax = amax_of(gy)
        if keep is not None:
            keep.append(gy_amax)
    if x_amax is None:                               # callers on the hot path pass the amax their producer left
        x_amax = act_amax_full(x, in_st) if in_st is not None else amax_of(x)
        if keep is not None:
            keep.append(x_amax)
    partial = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    dw = _dst(sink, (Cout, Cin, 3, 3), x.device)
    if keep is not None:
        keep.extend((partial, dw))                   # alive until the side stream has been joined
    with _timed("conv3x3_wgrad_sf16_mfma(+slice reduce)|%d->%d@%dx%d%s", (Cin, Cout, H, W, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        _call("sed_conv3x3_wgrad_sf16", _ptr(x), _ptr(gy), _ptr(dw), _ptr(partial), B, H, W, Cin, Cout,
              _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None,
              _ptr(gy_amax), _ptr(x_amax), _sf16_err_ptr(), _sf16_err_dev_ptr(x.device),
              (1 if x_presplit else 0) | (2 if gy_presplit else 0), _stream())
    return _ret(sink, dw) if signal else (None if sink is not None else dw)


WGRAD_SF16_MIN_CIN = int(os.environ.get("SED_WGRAD_SF16_MIN_CIN", "64"))


def _wgrad_algo(H, W, Cin, Cout):
    """3 = split-f16 (every layer it supports: Cin % 32 == 0.  Until round 3 the two 64-input-channel layers kept the
    Winograd-domain fp32 kernel -- the 64 x 32 tile reads every gradient row twice and only tied; with the 3-VALU split of
    round 3 it is 12-20 % faster there too), else the fp32 kernels."""
    return 3 if (USE_SF16 and Cin >= WGRAD_SF16_MIN_CIN and _lib.lib().sed_wgrad_sf16_supported(H, W, Cin, Cout)) else 0


def _wgrad(x, gy, B, H, W, Cin, Cout, in_st=None, sink=None, signal=True, gy_amax=None, x_amax=None, keep=None,
           x_presplit=False, gy_presplit=False):
    if _wgrad_algo(H, W, Cin, Cout) == 3:
        return _wgrad_sf16(x, gy, B, H, W, Cin, Cout, in_st=in_st, sink=sink, signal=signal, gy_amax=gy_amax, x_amax=x_amax,
                           keep=keep, x_presplit=x_presplit, gy_presplit=gy_presplit)
    if x_presplit or gy_presplit:
        raise RuntimeError("split-f16 operand pairs can only feed the split-f16 weight-gradient kernel")
    if USE_WINOGRAD >= 2 and W in (8, 16, 32, 64) and Cin % 32 == 0 and Cout % 64 == 0:
        return _wgrad_wino2(x, gy, B, H, W, Cin, Cout, in_st=in_st, sink=sink, signal=signal, keep=keep)
    return _wgrad_direct(x, gy, B, H, W, Cin, Cout, in_st=in_st, sink=sink, signal=signal, keep=keep)


def _wgrad_direct(x, gy, B, H, W, Cin, Cout, in_st=None, sink=None, signal=True, keep=None):
    ns, pps = ctypes.c_int(0), ctypes.c_int(0)
    nfl = _lib.lib().sed_wgrad_partial_floats(B * H * W, Cin, Cout, 9, ctypes.byref(ns), ctypes.byref(pps))
    partial = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    dw = _dst(sink, (Cout, Cin, 3, 3), x.device)
    if keep is not None:
        keep.extend((partial, dw))                   # alive until the side stream has been joined
    with _timed("conv3x3_wgrad_mfma(+slice reduce)|%d->%d@%dx%d%s", (Cin, Cout, H, W, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        _call("sed_conv3x3_wgrad", _ptr(x), _ptr(gy), _ptr(dw), _ptr(partial), B, H, W, Cin, Cout,
              _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None, _stream())
    return _ret(sink, dw) if signal else (None if sink is not None else dw)



# ---- split-f16 convolution (csrc/conv_sf16.hip)

AMAX_SLOTS = 64          # = sed_amax_slots(): device amax values are 64 floats (one atomic per producer block, spread over
                         # the slots; consumers take the max) -- see include/sed_hip.h
_AMAX_POOL = {}
_POOL_ROW = 128          # floats per pool row: an amax vector uses the first 64, a weight scale record 65


def _unregister_pool(addr, nfloats):
    try:
        _lib.lib().sed_amax_prezeroed_range(ctypes.c_void_p(addr), nfloats, 0)
    except Exception:                                   # interpreter shutdown
        pass


def _amax_buf(device, n=AMAX_SLOTS):
    """A ZEROED float[n] (n <= 128; default: a 64-slot amax vector): a slice of a pool that is zeroed 64 rows at a time
    (one fill kernel instead of one memset per producer launch).  The pool's ADDRESS RANGE is registered with the library
    (sed_amax_prezeroed_range) for as long as the pool's storage lives: only pointers inside it skip the library's own
    memset, every other amax buffer handed to the C ABI in this process is still zeroed by the entry point itself.  Each
    row is handed out once; the views keep their pool alive."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _AMAX_POOL.get(key)
    if ent is None or ent[1] >= ent[0].shape[0]:
        pool = torch.zeros((64, _POOL_ROW), dtype=torch.float32, device=torch.device("cuda", key))
        # the registry is a fixed table (64 live pools): when it is full -- many forward passes before one backward, retained
        # autograd graphs keep rows and hence pools alive -- the pool simply stays unregistered and the library zeroes its
        # rows itself, as it does for any other amax buffer (registration only SKIPS a memset, it is never needed)
        if _lib.lib().sed_amax_prezeroed_range(ctypes.c_void_p(pool.data_ptr()), pool.numel(), 1) == 0:
            weakref.finalize(pool, _unregister_pool, pool.data_ptr(), pool.numel())     # views keep `pool` (their _base) alive
        ent = _AMAX_POOL[key] = [pool, 0]
    v = ent[0][ent[1]][:n]
    ent[1] += 1
    return v


def reset_amax_pool():
    """The next _amax_buf() starts a fresh pool.  graph.GraphedTrainStep brackets its capture with this: rows handed out
    inside a captured region must come from a pool whose zero fill is PART of that region (a replay re-zeroes them), and
    eager code must never be handed rows of a pool a graph owns."""
    _AMAX_POOL.clear()


def amax_value(a):
    """Host float of a device amax vector (tests / debugging; synchronises)."""
    return float(a.max())


def pack_sf16(w_oihw, dgrad=False, wscale=None, both=False):
    """OIHW fp32 weights -> (split-f16 operand [hi, lo planes], wscale[65] = (64 amax slots, power-of-two scale) on the
    device).  wscale given: the amax pass is skipped (the second layout of a weight whose amax is already known).
    both: ONE launch writes the forward and the dgrad layout; returns ((fwd, wscale), (dgrad, wscale))."""
    Cout, Cin = w_oihw.shape[0], w_oihw.shape[1]
    halfs = _lib.lib().sed_conv_sf16_pack_halfs(Cin, Cout)
    wp = torch.empty((2 * halfs if both else halfs,), dtype=torch.float16, device=w_oihw.device)
    have = wscale is not None
    if not have:
        wscale = _amax_buf(w_oihw.device, AMAX_SLOTS + 1)
    _call("sed_pack_conv_weights_sf16", _ptr(_f32c(w_oihw)), Cout, Cin, (4 if both else (1 if dgrad else 0)) | (2 if have else 0),
          _ptr(wscale), _ptr(wp), _stream())
    if both:
        return (wp[:halfs], wscale), (wp[halfs:], wscale)
    return wp, wscale


def sf16_packs(w, want_dgrad):
    """(forward pack, dgrad pack or None) of a conv weight, cached per parameter object and parameter state (the same
    stamp as the other derived weight operands: storage address, version counter, optimiser generation): one amax pass
    and ONE pack launch per weight and optimiser step instead of one of each per use; static weights (inference) are
    packed once."""
    def build():
        if want_dgrad and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0:
            return list(pack_sf16(w, both=True))
        return [pack_sf16(w, dgrad=False), None]
    ent = _cached("sf16_packs", (w,), build)
    if want_dgrad and ent[1] is None:
        ent[1] = pack_sf16(w, dgrad=True, wscale=ent[0][1])
    return ent[0], ent[1]


def prepack_sf16(weights, want_dgrad):
    """The split-f16 operands of ALL the conv weights of a model whose cached packs are stale, in two launches
    (sed_pack_conv_weights_sf16_multi) instead of two per weight; leaves the same cache entries sf16_packs() builds, so
    the convolutions that follow find them.  Called by the models' trunk once per forward pass (a no-op while the
    parameters have not changed: inference, or a second forward pass before the optimiser step)."""
    todo = [w for w in weights if not _cache_fresh("sf16_packs", (w,))]
    if not todo:
        return 0
    L = _lib.lib()
    for i in range(0, len(todo), 16):
        grp = todo[i:i + 16]
        n = len(grp)
        both = [bool(want_dgrad) and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0 for w in grp]
        halfs = [int(L.sed_conv_sf16_pack_halfs(w.shape[1], w.shape[0])) for w in grp]
        wps = [torch.empty((2 * h if b else h,), dtype=torch.float16, device=w.device) for w, h, b in zip(grp, halfs, both)]
        wss = [_amax_buf(w.device, AMAX_SLOTS + 1) for w in grp]
        srcs = [_f32c(w) for w in grp]
        _call("sed_pack_conv_weights_sf16_multi", n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs]),
              (ctypes.c_int * n)(*[w.shape[0] for w in grp]), (ctypes.c_int * n)(*[w.shape[1] for w in grp]),
              (ctypes.c_int * n)(*[4 if b else 0 for b in both]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in wss]),
              (ctypes.c_void_p * n)(*[t.data_ptr() for t in wps]), _stream())
        for w, wp, ws, h, b in zip(grp, wps, wss, halfs, both):
            _cache_put("sf16_packs", (w,), [(wp[:h], ws), (wp[h:], ws)] if b else [(wp, ws), None])
    return len(todo)


def amax_of(x):
    """max |x| as a device amax vector (one pass)."""
    out = _amax_buf(x.device)
    _call("sed_amax", _ptr(x), x.numel(), _ptr(out), _stream())
    return out


def act_amax(minmax, nparts, C, st=None):
    """amax of relu(scale*y + shift) (st given) or of |y| from the per-part per-channel (max, min) a conv epilogue left --
    the split-f16 scale of a convolution whose operand is never materialised.  No pass over the tensor."""
    out = _amax_buf(minmax.device)
    _call("sed_act_amax", _ptr(minmax), nparts, C, _ptr(st.scale) if st is not None else None,
          _ptr(st.shift) if st is not None else None, _ptr(out), _stream())
    return out


def act_amax_full(y, st):
    """The same amax by one pass over y (producers that leave no range partials; off the default path)."""
    C = y.shape[-1]
    out = _amax_buf(y.device)
    _call("sed_act_amax_full", _ptr(y), y.numel() // C, C, _ptr(st.scale), _ptr(st.shift), _ptr(out), _stream())
    return out


def conv3x3_sf16(x, pack, B, H, W, Cin, Cout, in_st=None, epi=0, partials=None, yprev=None, p_st=None, x_amax=None,
                 minmax=None, presplit=False, out_amax=None):
    """y = conv3x3(relu(scale*x + shift) or x) on the f16 MFMA pipe with split operands; x NHWC fp32 -> y NHWC fp32.
    x_amax: device scalar = amax of the operand as the MFMAs see it (None: computed here by a pass over x);
    minmax: optional [nparts][2][Cout] buffer that receives the per-part range of y (for act_amax of the next conv)."""
    wp, wscale = pack
    if presplit and (x_amax is None or in_st is not None):
        raise RuntimeError("a pre-split operand comes with the amax it was scaled by and takes no input transform")
    if x_amax is None:
        x_amax = act_amax_full(x, in_st) if in_st is not None else amax_of(x)
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    ks, nfull = _conv_split_plan(B, H, W, Cin, Cout)
    with _timed("conv3x3_sf16_mfma(fwd+dgrad)|%d->%d@%dx%d epi%d%s", (Cin, Cout, H, W, epi, "+inT" if in_st is not None else ""),
                2.0 * 9 * B * H * W * Cin * Cout):
        args = (_ptr(x), _ptr(wp), _ptr(wscale), _ptr(y), B, H, W, Cin, Cout,
                _ptr(in_st.scale) if in_st is not None else None, _ptr(in_st.shift) if in_st is not None else None, epi,
                _ptr(partials), _ptr(yprev), _ptr(p_st.scale) if p_st is not None else None,
                _ptr(p_st.shift) if p_st is not None else None, _ptr(p_st.mean) if p_st is not None else None,
                _ptr(p_st.invstd) if p_st is not None else None, _ptr(x_amax), _ptr(minmax), _sf16_err_ptr(),
                _sf16_err_dev_ptr(x.device), 1 if presplit else 0, _ptr(out_amax))
        if ks > 1:
            # `ks` workgroups per output tile, each over 1/ks of the K-steps: every tile of a small-M launch (fewer workgroups than
            # resident slots; nfull = 0), or only the tiles of the last, partial round of the chip (nfull = the full rounds' tiles)
            L = _lib.lib()
            ws = torch.empty((L.sed_conv_sf16_splitk_floats(B, H, W, Cout, ks, nfull),), dtype=torch.float32, device=x.device)
            tickets = _splitk_tickets(x.device, L.sed_conv_sf16_splitk_tickets(B, H, W, Cout, nfull))
            _call("sed_conv3x3_sf16_splitk", *(args + (ks, nfull, _ptr(ws), _ptr(tickets), _stream())))
        else:
            _call("sed_conv3x3_sf16", *(args + (_stream(),)))
    return y


CONV_SPLITK = os.environ.get("SED_CONV_SPLITK", "1") != "0"      # convolutions may split the K range of (some of) their tiles over workgroups
# tail split of launches with a partial last round: 0 = the library's rule (never: measured slower, profiles/r06/tail_split_ab.txt),
# 2 .. 8 = force that many shares per tail tile (A/B runs: SED_CONV_TAIL; tests set it directly)
CONV_TAIL = int(os.environ.get("SED_CONV_TAIL", "0") or 0)
_KSPLIT = {}
_TICKETS = {}


def _conv_split_plan(B, H, W, Cin, Cout):
    """(ksplit, nfull) of a split-f16 convolution launch: (1, 0) = un-split; nfull = 0 = every tile split (small-M launches);
    nfull > 0 = only the tiles behind the first nfull -- the last, partial round of the chip -- are split (csrc/conv_sf16.hip)."""
    if not CONV_SPLITK:
        return 1, 0
    key = (B, H, W, Cin, Cout, CONV_TAIL)
    v = _KSPLIT.get(key)
    if v is None:
        nfull = ctypes.c_int(0)
        ks = int(_lib.lib().sed_conv_sf16_split_plan(B, H, W, Cin, Cout, CONV_TAIL, ctypes.byref(nfull)))
        v = _KSPLIT[key] = (ks, int(nfull.value) if ks > 1 else 0)
    return v


def _splitk_tickets(device, n):
    """Ticket words of the split-K convolutions, one buffer per (device, stream): zero once; every launch leaves them zero (the
    last workgroup of a tile resets its ticket) and launches on ONE stream never overlap -- two streams must not share tickets."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(_stream() or 0))
    t = _TICKETS.get(key)
    if t is None or t.numel() < n:
        t = _TICKETS[key] = torch.zeros((max(int(n), 4096),), dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def _conv_algo(H, W, Cin, Cout):
    """3 = split-f16 direct, 2 = fused 2-D Winograd F(2x2,3x3) on fp32 MFMA, 0 = direct fp32 implicit GEMM."""
    L = _lib.lib()
    if USE_SF16 and L.sed_conv3x3_sf16_supported(H, W, Cin, Cout):
        return 3
    if USE_WINOGRAD >= 2 and L.sed_conv3x3_wino2_supported(H, W, Cin, Cout):
        return 2
    return 0


def _conv_fwd_like(x, w_oihw, B, H, W, Cin, Cout, dgrad=False, **kw):
    """3x3 conv of x with the OIHW weights (dgrad=True: the transposed/flipped conv that maps g_y -> g_x; Cin/Cout are
    then the channel counts of the INPUT/OUTPUT of this call)."""
    algo = _conv_algo(H, W, Cin, Cout)
    x_amax = kw.pop("x_amax", None)
    minmax = kw.pop("minmax", None)
    packs = kw.pop("packs", None)                    # (forward, dgrad) split-f16 packs of w_oihw, when the caller holds them
    presplit = kw.pop("presplit", False)
    out_amax = kw.pop("out_amax", None)
    if algo == 3:
        pack = packs[1 if dgrad else 0] if packs is not None else None
        if pack is None:
            pack = sf16_packs(w_oihw, dgrad)[1 if dgrad else 0]
        return conv3x3_sf16(x, pack, B, H, W, Cin, Cout, x_amax=x_amax, minmax=minmax, presplit=presplit, out_amax=out_amax, **kw)
    if presplit or out_amax is not None:
        raise RuntimeError("split-f16 operand pairs can only feed the split-f16 convolution kernel")
    if algo == 2:
        uf, ud = _pack_wino2(w_oihw, want_f=not dgrad, want_d=dgrad)
        return _conv_wino2(x, ud if dgrad else uf, B, H, W, Cin, Cout, **kw)
    wf, wd = _pack(w_oihw, want_f=not dgrad, want_d=dgrad)
    return _conv_igemm(x, wd if dgrad else wf, B, H, W, Cin, Cout, **kw)


def _conv_parts(B, H, W, Cin, Cout):
    """(number of statistics partials, rows per partial [-1: counts appended], floats to allocate) written by
    _conv_fwd_like for this shape."""
    L = _lib.lib()
    M = B * H * W
    algo = _conv_algo(H, W, Cin, Cout)
    if algo == 3:
        P = int(L.sed_conv_sf16_num_parts(B, H, W, Cout))
        return P, -1, P * 2 * Cout + P
    if algo == 2:
        P = int(L.sed_conv_wino2_num_parts(B, H, W))
        return P, -1, P * 2 * Cout + P
    P = L.sed_conv_num_parts(M, Cout)
    return P, L.sed_conv_rows_per_part(M, Cout), P * 2 * Cout


def block_out_pairs_ok(training, pool_mode, ph, pw, H, W, Cout):
    """Will ConvBlockFn(..., out_pairs=True) really write its pooled output as operand pairs?  (Asked by the module, which
    must tell the next block; decided by the same rule inside the Function.  Both add "a backward pass will follow": the module
    from torch.is_grad_enabled(), the Function -- whose forward always runs with grad mode off -- from its no_backward flag.)"""
    return bool(ACT_PAIRS and GRAD_PAIRS and USE_SF16 and training and pool_mode == 0 and ph * pw > 1
                and POOL_BWD_WINDOWED and _conv_algo(H, W, Cout, Cout) == 3 and _wgrad_algo(H, W, Cout, Cout) == 3
                and _conv_algo(H // ph, W // pw, Cout, 2 * Cout) == 3 and _wgrad_algo(H // ph, W // pw, Cout, 2 * Cout) == 3)


class ConvBlockFn(torch.autograd.Function):
    """One reference ConvBlock (models.py:99-115, pool_type='avg'), NHWC, training or eval.
    x (B,H,W,Cin) -> (B,H//ph,W//pw,Cout).  Only the two raw conv outputs are saved; BN+ReLU is recomputed
    on the fly by the consumers."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, training, ph, pw, x_amax=None, pool_mode=0, no_backward=False,
                x_pairs=False, out_pairs=False):
        """Returns (out, out_amax): out_amax = device amax vector of `out` for the next block's split-f16 scale (x_amax
        there).  pool_mode: 0 = 'avg' (every model), 1 = 'max', 2 = 'avg+max' (models.py:104-111).  no_backward: the caller
        runs under torch.no_grad() (the Function cannot see that itself): nothing is kept for a backward pass and the
        inference epilogue may be used."""
        _chk_dev(x, w1, w2)
        ctx.set_materialize_grads(False)     # out_amax takes no gradient: spare autograd its zero tensor (a fill launch per block)
        x = _f32c(x)
        B, H, W, Cin = x.shape
        Cout = w1.shape[0]
        dev = x.device
        L = _lib.lib()
        M = B * H * W
        w1c, w2c = _f32c(w1), _f32c(w2)
        # split-f16 operand scales come from the amax of each operand, left on the device by its producer
        # x_pairs: x holds split-f16 operand pairs (the previous block's pooled output, scaled by x_amax); out_pairs: write ours so
        sf_c1 = Cin != 1 and _conv_algo(H, W, Cin, Cout) == 3
        if x_pairs and not (sf_c1 and x_amax is not None and (not training or _wgrad_algo(H, W, Cin, Cout) == 3)):
            raise RuntimeError("operand pairs can only feed the split-f16 kernels (and come with the amax they were scaled by)")
        out_pairs = bool(out_pairs) and not no_backward and block_out_pairs_ok(training, pool_mode, ph, pw, H, W, Cout)
        need_xa = sf_c1 or (training and Cin != 1 and _wgrad_algo(H, W, Cin, Cout) == 3)
        need_a1 = _conv_algo(H, W, Cout, Cout) == 3 or (training and _wgrad_algo(H, W, Cout, Cout) == 3)
        if need_xa and x_amax is None:
            x_amax = amax_of(x)                          # standalone use; inside a model the previous block's pool supplies it
        if not need_xa:
            x_amax = None
        # split-f16 weight operands: one amax pass + packs per weight and optimiser step (cached per parameter), both
        # layouts at once when a backward pass will follow
        pk1 = sf16_packs(w1c, bool(training) and ctx.needs_input_grad[0]) if sf_c1 else None
        pk2 = sf16_packs(w2c, bool(training)) if _conv_algo(H, W, Cout, Cout) == 3 else None
        # conv1 (+ statistics, + per-channel output range for the amax of relu(bn1(y1)))
        mm1 = None
        keep_mm1 = None
        b1_pairs = (B1_ACT_PAIRS and Cin == 1 and training and not no_backward and USE_SF16 and pool_mode == 0
                    and _conv_algo(H, W, Cout, Cout) == 3 and _wgrad_algo(H, W, Cout, Cout) == 3 and pk2 is not None
                    and pk2[1] is not None)
        if b1_pairs:
            rpp1 = L.sed_conv1_rows_per_part()
            np1 = (M + rpp1 - 1) // rpp1
            part1 = torch.empty((np1, 2, Cout), dtype=torch.float32, device=dev)
            mm1 = torch.empty((np1, 2, Cout), dtype=torch.float32, device=dev)
            _call("sed_conv1_fwd", _ptr(x), _ptr(w1c), None, B, H, W, _ptr(part1), _ptr(mm1), _stream())   # statistics + range only
            a1 = _amax_buf(dev)
            st1 = bn_finalize(part1, np1, rpp1, M, g1, b1, rm1, rv1, minmax=mm1, act_amax_out=a1)
            del mm1
            y1 = torch.empty((B, H, W, Cout), dtype=torch.int32, device=dev)       # split-f16 PAIRS of relu(bn1(conv1(x)))
            _call("sed_conv1_act_sf16", _ptr(x), _ptr(w1c), B, H, W, _ptr(st1.scale), _ptr(st1.shift), _ptr(a1), _ptr(y1),
                  _sf16_err_ptr(), _sf16_err_dev_ptr(dev), _stream())
        elif Cin == 1:
            rpp1 = L.sed_conv1_rows_per_part()
            np1 = (M + rpp1 - 1) // rpp1
            part1 = torch.empty((np1, 2, Cout), dtype=torch.float32, device=dev) if training else None
            mm1 = torch.empty((np1, 2, Cout), dtype=torch.float32, device=dev) if need_a1 else None
            y1 = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
            _call("sed_conv1_fwd", _ptr(x), _ptr(w1c), _ptr(y1), B, H, W, _ptr(part1), _ptr(mm1), _stream())
        else:
            np1, rpp1, nf1 = _conv_parts(B, H, W, Cin, Cout)
            part1 = torch.empty((nf1,), dtype=torch.float32, device=dev) if training else None
            if need_a1 and sf_c1:
                mm1 = torch.empty((np1, 2, Cout), dtype=torch.float32, device=dev)
                keep_mm1 = (mm1, np1)                    # backward: the range of y1 bounds bn1's backward output (GRAD_PAIRS)
            y1 = _conv_fwd_like(x, w1c, B, H, W, Cin, Cout, epi=1 if training else 0, partials=part1, x_amax=x_amax, minmax=mm1,
                                packs=pk1, presplit=bool(x_pairs))
        if not b1_pairs:
            a1 = None
            if training and need_a1 and mm1 is not None:
                # the operand amax of conv2 -- relu(bn1(y1)), never materialised -- from the range conv1's epilogue left, by the
                # finalize launch itself (round 6: one launch fewer per block)
                a1 = _amax_buf(dev)
                st1 = bn_finalize(part1, np1, rpp1, M, g1, b1, rm1, rv1, minmax=mm1, act_amax_out=a1)
            else:
                st1 = bn_finalize(part1, np1, rpp1, M, g1, b1, rm1, rv1) if training else bn_eval_affine(g1, b1, rm1, rv1)
                if need_a1:
                    a1 = act_amax(mm1, np1, Cout, st1) if mm1 is not None else act_amax_full(y1, st1)
            del mm1
        if (EVAL_POOL_FUSION and no_backward and not training and pool_mode == 0 and pk2 is not None
                and L.sed_conv3x3_sf16_eval_pool_supported(H, W, Cout, Cout, ph, pw)):
            # inference: conv2 + eval-mode BatchNorm + ReLU + average pool in ONE kernel -- the
            # full-resolution y2 is never written, the pool pass never runs
            st2 = bn_eval_affine(g2, b2, rm2, rv2)
            out = torch.empty((B, H // ph, W // pw, Cout), dtype=torch.float32, device=dev)
            out_amax = _amax_buf(dev)
            _call("sed_conv3x3_sf16_eval_pool", _ptr(y1), _ptr(pk2[0][0]), _ptr(pk2[0][1]), _ptr(out), B, H, W, Cout, Cout,
                  _ptr(st1.scale), _ptr(st1.shift), _ptr(st2.scale), _ptr(st2.shift), ph, pw, _ptr(a1), _ptr(out_amax),
                  _sf16_err_ptr(), _sf16_err_dev_ptr(dev), _stream())
            ctx.mark_non_differentiable(out_amax)
            return out, out_amax
        # conv2 over relu(bn1(y1)) computed on the fly (+ statistics)
        np2, rpp2, nf2 = _conv_parts(B, H, W, Cout, Cout)
        part2 = torch.empty((nf2,), dtype=torch.float32, device=dev) if training else None
        keep_mm2 = None
        if (GRAD_PAIRS and training and not no_backward and pool_mode == 0 and pk2 is not None
                and _wgrad_algo(H, W, Cout, Cout) == 3):
            keep_mm2 = _amax_buf(dev)        # amax of y2, published by conv2's epilogue: bounds bn2's backward output
        y2 = _conv_fwd_like(y1, w2c, B, H, W, Cout, Cout, in_st=None if b1_pairs else st1, epi=1 if training else 0,
                            partials=part2, x_amax=a1, packs=pk2, presplit=b1_pairs, out_amax=keep_mm2)
        out = torch.empty((B, H // ph, W // pw, Cout), dtype=torch.float32, device=dev)
        out_amax = _amax_buf(dev)
        pairs_now = pool_mode == 0 and out_pairs and keep_mm2 is not None
        # (pairs: the pooled output is scaled by a bound of its amax known before the pass -- from the amax of y2 and bn2's affine,
        # computed by the finalize launch itself)
        st2 = bn_finalize(part2, np2, rpp2, M, g2, b2, rm2, rv2, y_amax=keep_mm2 if pairs_now else None,
                          act_bound_out=out_amax if pairs_now else None) if training else bn_eval_affine(g2, b2, rm2, rv2)
        cnt = None
        if pool_mode != 0:
            _call("sed_bn_relu_pool_fwd_mode", _ptr(y2), B, H, W, Cout, ph, pw, int(pool_mode), _ptr(st2.scale), _ptr(st2.shift),
                  _ptr(out), _ptr(out_amax), _stream())
        elif out_pairs and keep_mm2 is not None:
            # the pooled output as split-f16 operand pairs, scaled by a bound of its amax known before the pass
            cnt = torch.empty((B, H // ph, W // pw, Cout), dtype=torch.uint8, device=dev)
            _call("sed_bn_relu_pool_fwd_cnt_pairs", _ptr(y2), B, H, W, Cout, ph, pw, _ptr(st2.scale), _ptr(st2.shift), _ptr(out),
                  _ptr(cnt), _ptr(out_amax), _stream())
        elif training and POOL_BWD_WINDOWED and ph * pw > 1:
            # per-window ReLU counts: with them backward pass 1 runs on the pooled tensors and never reads y2
            cnt = torch.empty((B, H // ph, W // pw, Cout), dtype=torch.uint8, device=dev)
            _call("sed_bn_relu_pool_fwd_cnt", _ptr(y2), B, H, W, Cout, ph, pw, _ptr(st2.scale), _ptr(st2.shift), _ptr(out),
                  _ptr(cnt), _ptr(out_amax), _stream())
        else:
            _call("sed_bn_relu_pool_fwd", _ptr(y2), B, H, W, Cout, ph, pw, _ptr(st2.scale), _ptr(st2.shift), _ptr(out),
                  _ptr(out_amax), _stream())
        if cnt is not None:
            ctx.save_for_backward(x, y1, y2, w1c, w2c, out, cnt, _f32c(g2), _f32c(b2))
        else:
            ctx.save_for_backward(x, y1, y2, w1c, w2c)
        ctx.b1_pairs = b1_pairs
        ctx.x_pairs = bool(x_pairs)
        ctx.out_bound = out_amax if (out_pairs and keep_mm2 is not None) else None
        ctx.mm1, ctx.mm2 = (keep_mm1 if GRAD_PAIRS and training else None), keep_mm2
        ctx.st1, ctx.st2, ctx.pool, ctx.training, ctx.pool_mode = st1, st2, (ph, pw), bool(training), int(pool_mode)
        ctx.xa, ctx.a1 = x_amax, a1
        ctx.pk1, ctx.pk2 = pk1, pk2
        ctx.sinks = _sinks(ctx, (w1, g1, b1, None, None, w2, g2, b2), 1)
        ctx.mark_non_differentiable(out_amax)
        return out, out_amax

    @staticmethod
    def backward(ctx, g_out, _g_amax=None):
        x, y1, y2, w1, w2 = ctx.saved_tensors[:5]
        win = ctx.saved_tensors[5:] if len(ctx.saved_tensors) > 5 else None
        st1, st2 = ctx.st1, ctx.st2
        ph, pw = ctx.pool
        g_out = _f32c(g_out)
        B, H, W, Cin = x.shape
        Cout = w1.shape[0]
        dev = x.device
        M = B * H * W
        # BN2 + ReLU + pool backward
        n = ctypes.c_int(0)
        if win is not None:
            pooled, cnt, gam, bet = win
            npmax = _lib.lib().sed_bn_relu_pool_bwd_reduce_auto_parts(B, H, W, ph, pw)
            part = torch.empty((npmax, 2, Cout), dtype=torch.float32, device=dev)
            _call("sed_bn_relu_pool_bwd_reduce_auto", _ptr(y2), _ptr(g_out), _ptr(pooled), _ptr(cnt), B, H, W, Cout, ph, pw,
                  _ptr(st2.scale), _ptr(st2.shift), _ptr(st2.mean), _ptr(st2.invstd), _ptr(gam), _ptr(bet), POOL_BWD_GAMMA_MIN,
                  _ptr(part), ctypes.byref(n), _ptr(ctx.out_bound), _stream())
            del pooled, cnt, win
        else:
            rpb = _lib.lib().sed_pool_bwd_rows_per_block(M)
            npmax = (M + rpb - 1) // rpb
            part = torch.empty((npmax, 2, Cout), dtype=torch.float32, device=dev)
            _call("sed_bn_relu_pool_bwd_reduce_mode", _ptr(y2), _ptr(g_out), B, H, W, Cout, ph, pw, ctx.pool_mode, _ptr(st2.scale),
                  _ptr(st2.shift), _ptr(st2.mean), _ptr(st2.invstd), _ptr(part), ctypes.byref(n), _stream())
        sk = ctx.sinks                                   # (w1, g1, b1, -, -, w2, g2, b2)
        sf2 = _conv_algo(H, W, Cout, Cout) == 3 or _wgrad_algo(H, W, Cout, Cout) == 3   # split-f16 consumers scale by the amax
        # gradients as operand pairs: both consumers of gy2 (conv2's dgrad and weight gradient) must be the split-f16 kernels
        pair2 = (ctx.mm2 is not None and ctx.pool_mode == 0 and _conv_algo(H, W, Cout, Cout) == 3
                 and _wgrad_algo(H, W, Cout, Cout) == 3)
        if pair2:
            ent = _GRAD_AMAX.pop(g_out.data_ptr(), None)
            g_amax = ent[0] if (ent is not None and ent[1] == g_out.numel()) else amax_of(g_out)
            amax2 = _amax_buf(dev)                       # an upper BOUND of max |gy2|: the scale the pairs are written with,
            dg2, db2, coef2 = bn_bwd_finalize(part, n.value, M, st2, batch_stats=ctx.training, sinks=(sk[6], sk[7]),   # left by the
                                              bound=(ctx.mm2, g_amax, 1.0 / float(ph * pw), amax2))                   # finalize launch
        else:
            dg2, db2, coef2 = bn_bwd_finalize(part, n.value, M, st2, batch_stats=ctx.training, sinks=(sk[6], sk[7]))
        gy2 = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
        if pair2:
            _call("sed_bn_relu_pool_bwd_apply_pairs", _ptr(y2), _ptr(g_out), B, H, W, Cout, ph, pw, _ptr(st2.scale), _ptr(st2.shift),
                  _ptr(coef2), _ptr(gy2), _ptr(amax2), _stream())
        else:
            amax2 = _amax_buf(dev) if sf2 else None
            _call("sed_bn_relu_pool_bwd_apply_mode", _ptr(y2), _ptr(g_out), B, H, W, Cout, ph, pw, ctx.pool_mode, _ptr(st2.scale),
                  _ptr(st2.shift), _ptr(coef2), _ptr(gy2), _ptr(amax2), _stream())
        # will gy1 (bn1's backward output) be written as pairs?  Its consumers: conv1's dgrad and weight gradient (Cin != 1)
        pair1 = (Cin != 1 and ctx.mm1 is not None and _wgrad_algo(H, W, Cin, Cout) == 3
                 and (not ctx.needs_input_grad[0] or _conv_algo(H, W, Cout, Cin) == 3))
        d_amax = _amax_buf(dev) if pair1 else None       # amax of conv2's (masked) dgrad output, published by its epilogue
        # conv2: dgrad fused with relu-mask + BN1 backward sums, then the weight gradient (operand relu(bn1(y1)) on the
        # fly).  With gradient sinks the weight gradients run on the side stream: conv2's beside BN1's backward passes
        # below, conv1's beside the NEXT block's pool backward (joined there, right here, before its first MFMA kernel).
        join_side_stream()
        fork = WGRAD_SIDE_STREAM
        npb, _, nfb = _conv_parts(B, H, W, Cout, Cout)
        partb = torch.empty((nfb,), dtype=torch.float32, device=dev)
        b1_pairs = ctx.b1_pairs
        if b1_pairs:
            # block 1: y1 holds the split-f16 pairs of a1 = relu(bn1(conv1(x))); the raw conv1 output the ReLU mask and xhat
            # need is recomputed from x inside the kernel's epilogue (bit-identical to the tensor it used to read back)
            gy1 = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
            wpd, wsd = ctx.pk2[1]
            with _timed("conv3x3_sf16_mfma(fwd+dgrad)|%d->%d@%dx%d epi4", (Cout, Cout, H, W), 2.0 * 9 * B * H * W * Cout * Cout):
                _call("sed_conv3x3_sf16_dgrad_b1", _ptr(gy2), _ptr(wpd), _ptr(wsd), _ptr(gy1), B, H, W, Cout, Cout, _ptr(partb),
                      _ptr(st1.scale), _ptr(st1.shift), _ptr(st1.mean), _ptr(st1.invstd), _ptr(x), _ptr(w1), _ptr(amax2),
                      _sf16_err_ptr(), _sf16_err_dev_ptr(dev), 1 if pair2 else 0, None, _stream())
        else:
            gy1 = _conv_fwd_like(gy2, w2, B, H, W, Cout, Cout, dgrad=True, epi=2, partials=partb, yprev=y1, p_st=st1, x_amax=amax2,
                                 packs=ctx.pk2, presplit=pair2, out_amax=d_amax)
        w_in_st = None if b1_pairs else st1
        if fork and sk[5] is not None:
            dw2 = _fork_wgrad(y1, gy2, B, H, W, Cout, Cout, in_st=w_in_st, sink=sk[5], gy_amax=amax2, x_amax=ctx.a1, x_presplit=b1_pairs,
                              gy_presplit=pair2)
        else:
            dw2 = _wgrad(y1, gy2, B, H, W, Cout, Cout, in_st=w_in_st, sink=sk[5], gy_amax=amax2, x_amax=ctx.a1, x_presplit=b1_pairs,
                         gy_presplit=pair2)
        del gy2
        amax1 = None
        if Cin != 1 and pair1 and ctx.mm1[1] == npb:
            # BOUND of max |a*gy1 + b*y1 + c| from the range of y1 and the amax of gy1, left by the finalize launch itself (round 6;
            # the forward convolution and this dgrad cut the tensor into the same parts)
            amax1 = _amax_buf(dev)
            dg1, db1, coef1 = bn_bwd_finalize(partb, npb, M, st1, batch_stats=ctx.training, sinks=(sk[1], sk[2]),
                                              bound=(None, d_amax, 1.0, amax1), minmax=ctx.mm1[0])
        else:
            dg1, db1, coef1 = bn_bwd_finalize(partb, npb, M, st1, batch_stats=ctx.training, sinks=(sk[1], sk[2]))
        # conv1
        gx = None
        if Cin != 1 and pair1:
            if amax1 is None:
                amax1 = _amax_buf(dev)
                _call("sed_grad_bound", _ptr(ctx.mm1[0]), ctx.mm1[1], Cout, _ptr(coef1), _ptr(d_amax), 1.0, _ptr(amax1), None, _stream())
            _call("sed_bn_bwd_apply_pairs", _ptr(gy1), _ptr(y1), M, Cout, _ptr(coef1), _ptr(amax1), _stream())
        elif Cin != 1:
            if (ctx.needs_input_grad[0] and _conv_algo(H, W, Cout, Cin) == 3) or _wgrad_algo(H, W, Cin, Cout) == 3:
                amax1 = _amax_buf(dev)
            _call("sed_bn_bwd_apply", _ptr(gy1), _ptr(y1), M, Cout, _ptr(coef1), _ptr(amax1), _stream())
        if Cin == 1:                                   # BN1 backward g = a*dz + b*y1 + c is applied on load by the kernel
            dwp = torch.empty((int(_lib.lib().sed_conv1_bwd_partial_floats(B, H, W)),), dtype=torch.float32, device=dev)
            dw1 = _dst(sk[0], (Cout, 1, 3, 3), dev)
            want_gx = ctx.needs_input_grad[0]
            tbuf = torch.empty((M, 9), dtype=torch.float32, device=dev) if want_gx else None
            gx = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev) if want_gx else None
            # (block 1 without a materialised y1: the kernel recomputes it from x)
            _call("sed_conv1_bwd", _ptr(x), _ptr(w1), _ptr(gy1), None if b1_pairs else _ptr(y1), _ptr(coef1), B, H, W, _ptr(dw1),
                  _ptr(gx), _ptr(dwp), _ptr(tbuf), _stream())
            dw1 = _ret(sk[0], dw1)
            join_side_stream()
        else:
            join_side_stream()                         # conv2's weight gradient is done before the next MFMA kernel starts
            if ctx.needs_input_grad[0]:
                gx_amax = _amax_buf(dev) if (GRAD_PAIRS and _conv_algo(H, W, Cout, Cin) == 3) else None
                gx = _conv_fwd_like(gy1, w1, B, H, W, Cout, Cin, dgrad=True, epi=0, x_amax=amax1, packs=ctx.pk1, presplit=pair1,
                                    out_amax=gx_amax)
                if gx_amax is not None:                  # the previous block's backward bounds ITS gradient with it
                    if not _GRAD_AMAX:
                        torch.autograd.Variable._execution_engine.queue_callback(_GRAD_AMAX.clear)     # end of this backward pass
                    _GRAD_AMAX[gx.data_ptr()] = (gx_amax, gx.numel())
            if fork and sk[0] is not None:
                dw1 = _fork_wgrad(x, gy1, B, H, W, Cin, Cout, sink=sk[0], gy_amax=amax1, x_amax=ctx.xa, gy_presplit=pair1,
                                  x_presplit=ctx.x_pairs)
            else:
                dw1 = _wgrad(x, gy1, B, H, W, Cin, Cout, sink=sk[0], gy_amax=amax1, x_amax=ctx.xa, gy_presplit=pair1,
                             x_presplit=ctx.x_pairs)
        return gx, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------------
# dense helpers

def gemm_nt(x, w, bias=None):
    """y[M][N] = x[M][K] w[N][K]^T (+bias)."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _call("sed_gemm_nt", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, N, K, _stream())
    return y


# Dense layers on the f16 MFMA pipe with split-f16 operands (csrc/gemm_sf16.hip): the nn.GRU input projections and their input
# gradient.  Same arithmetic contract as the split-f16 convolutions (fp32-level error, scales from device-side amax values, a
# non-finite operand raises the guard words); falls back to the fp32 MFMA GEMM for shapes it does not take or with USE_SF16 off.
GEMM_SF16 = os.environ.get("SED_GEMM_SF16", "1") != "0"


def gemm_pack_sf16(w):
    """w [N][K] fp32 -> (split-f16 pack, wscale[65]) for gemm_nt_sf16 (two launches: amax, pack)."""
    N, K = w.shape
    wp = torch.empty((int(_lib.lib().sed_gemm_pack_sf16_halfs(N, K)),), dtype=torch.float16, device=w.device)
    ws = _amax_buf(w.device, AMAX_SLOTS + 1)
    _call("sed_gemm_pack_sf16", _ptr(w), N, K, _ptr(ws), _ptr(wp), _stream())
    return wp, ws


def gemm_nt_sf16_ok(M, N, K):
    return bool(GEMM_SF16 and USE_SF16 and _lib.lib().sed_gemm_nt_sf16_supported(M, N, K))


def gemm_nt_sf16(x, pack, N, bias=None, x_amax=None, out_amax=None):
    """y[M][N] = x[M][K] w[N][K]^T (+bias) with w given as its split-f16 pack; x_amax: device amax vector of x (None: one pass)."""
    M, K = x.shape
    wp, ws = pack
    if x_amax is None:
        x_amax = amax_of(x)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _timed("gemm_nt_sf16_mfma|%dx%dx%d", (M, N, K), 2.0 * M * N * K):
        _call("sed_gemm_nt_sf16", _ptr(x), _ptr(wp), _ptr(ws), _ptr(bias), _ptr(y), M, N, K, _ptr(x_amax), _sf16_err_ptr(),
              _sf16_err_dev_ptr(x.device), _ptr(out_amax), _stream())
    return y


def linear_nt(x, w, bias=None, transposed=False, x_amax=None):
    """y = x w^T (+ bias) of an nn.Linear weight w [N][K] -- or, transposed=True, y = x w: the input gradient of the same layer --
    on the split-f16 GEMM where the shape allows it (the pack of w / w^T is cached per parameter and optimiser step like the
    convolution packs), else on the fp32 MFMA GEMM."""
    M, K = x.shape
    N = w.shape[1] if transposed else w.shape[0]
    if gemm_nt_sf16_ok(M, N, K):
        def build():
            wc = _f32c(w.detach())
            if transposed:
                wc = transpose_b(wc.view(1, wc.shape[0], wc.shape[1])).view(wc.shape[1], wc.shape[0])
            return gemm_pack_sf16(wc)
        return gemm_nt_sf16(x, _cached("lin_sf16_t" if transposed else "lin_sf16", (w,), build), N, bias, x_amax=x_amax)
    wc = _f32c(w)
    if transposed:
        wc = transpose_b(wc.view(1, wc.shape[0], wc.shape[1])).view(wc.shape[1], wc.shape[0])
    return gemm_nt(x, wc, bias)


_UNIT_AMAX = {}


def unit_amax(device):
    """Device amax vector holding 1.0: the operand scale of tensors that are bounded by construction (GRU hidden states)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _UNIT_AMAX.get(key)
    if t is None:
        t = _UNIT_AMAX[key] = torch.ones((AMAX_SLOTS,), dtype=torch.float32, device=torch.device("cuda", key))
    return t


def gemm_tn(x, gy, out=None, x_amax=None, gy_amax=None):
    """dw[N][K] = sum_m gy[m][n] x[m][k]: on the split-f16 TN GEMM where the shape allows it (x_amax / gy_amax: device amax
    vectors or upper bounds of them; None = one pass each), else on the fp32 MFMA kernel."""
    M, K = x.shape
    N = gy.shape[1]
    if GEMM_SF16 and USE_SF16 and x.is_contiguous() and gy.is_contiguous() and _lib.lib().sed_gemm_tn_sf16_supported(M, N, K):
        if x_amax is None:
            x_amax = amax_of(x)
        if gy_amax is None:
            gy_amax = amax_of(gy)
        partial = torch.empty((int(_lib.lib().sed_gemm_tn_sf16_partial_floats(M, N, K)),), dtype=torch.float32, device=x.device)
        dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=x.device)
        with _timed("gemm_tn_sf16_mfma(+slice reduce)|%dx%dx%d", (M, N, K), 2.0 * M * N * K):
            _call("sed_gemm_tn_sf16", _ptr(x), _ptr(gy), _ptr(dw), _ptr(partial), M, N, K, _ptr(x_amax), _ptr(gy_amax),
                  _sf16_err_ptr(), _sf16_err_dev_ptr(x.device), _stream())
        return dw
    ns, pps = ctypes.c_int(0), ctypes.c_int(0)
    nfl = _lib.lib().sed_wgrad_partial_floats(M, K, N, 1, ctypes.byref(ns), ctypes.byref(pps))
    partial = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=x.device)
    _call("sed_gemm_tn", _ptr(x), _ptr(gy), _ptr(dw), _ptr(partial), M, N, K, _stream())
    return dw


def col_sums(x2d, ncols=None, out=None):
    n, ld = x2d.shape
    K = ld if ncols is None else ncols
    if out is None:
        out = torch.empty((K,), dtype=torch.float32, device=x2d.device)
    ws = torch.empty((256 * K,), dtype=torch.float32, device=x2d.device) if n > 512 else None
    _call("sed_reduce_rows", _ptr(x2d), n, K, ld, _ptr(out), 0, _ptr(ws), _stream())
    return out


def transpose_b(x):
    """(B, R, C) -> (B, C, R) contiguous."""
    B, R, C = x.shape
    out = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    _call("sed_transpose", _ptr(x), B, R, C, _ptr(out), _stream())
    return out


class InterpolateFn(torch.autograd.Function):
    """models.py:58-69.  (B,T,K) -> (B,T*ratio,K), each frame repeated `ratio` times.  The backward (sum over the
    repeats) only runs for a loss on `framewise_output`, which the reference's training never uses: off the hot path."""

    @staticmethod
    def forward(ctx, frame, ratio):
        _chk_dev(frame)
        frame = _f32c(frame)
        B, T, K = frame.shape
        out = torch.empty((B, T * ratio, K), dtype=torch.float32, device=frame.device)
        _call("sed_interpolate", _ptr(frame), B * T, K, ratio, _ptr(out), _stream())
        ctx.ratio = ratio
        return out

    @staticmethod
    def backward(ctx, g):
        B, TR, K = g.shape
        return g.reshape(B, TR // ctx.ratio, ctx.ratio, K).sum(dim=2), None


def interpolate(frame, ratio):
    return InterpolateFn.apply(frame, ratio)


LDN = 64   # padded logit width of the 17-class heads (MFMA GEMM N granularity)


def _pad_rows(ws, device):
    """Stack weight matrices (each (17, K)) into a zero-padded (64, K) operand and its (K, 64) transpose (device copies,
    cached until a source parameter changes)."""
    def build():
        K = ws[0].numel() // ws[0].shape[0]
        out = torch.zeros((LDN, K), dtype=torch.float32, device=device)
        r = 0
        for w in ws:
            w2 = w.detach().reshape(w.shape[0], K)
            out[r:r + w2.shape[0]].copy_(w2)
            r += w2.shape[0]
        return out, transpose_b(out.view(1, LDN, K)).view(K, LDN)
    return _cached("pad_rows", ws, build)


class FcHeadFn(torch.autograd.Function):
    """FrameAvg (mode 0, models.py:306-312) / FrameMax (mode 1, :221-227) head.
    feat (B,T,512) -> frame (B,T,17), clip (B,17).  The training loss (clip_bce) feeds `clip`; a gradient arriving
    through `frame` (a strong-label loss, not used by the reference's main.py) is honoured too, off the hot path."""

    @staticmethod
    def forward(ctx, feat, w, b, mode):
        _chk_dev(feat, w)
        feat = _f32c(feat)
        B, T, C = feat.shape
        ncls = w.shape[0]
        wp, wpt = _pad_rows([w], feat.device)
        logits = gemm_nt(feat.view(B * T, C), wp)
        frame = torch.empty((B, T, ncls), dtype=torch.float32, device=feat.device)
        clip = torch.empty((B, ncls), dtype=torch.float32, device=feat.device)
        amax = torch.empty((B, ncls), dtype=torch.int32, device=feat.device) if mode == 1 else None
        _call("sed_head_pool_fwd", _ptr(logits), B, T, LDN, ncls, _ptr(_f32c(b)), mode, _ptr(frame), _ptr(clip), _ptr(amax),
              _stream())
        ctx.save_for_backward(feat, wpt, frame, amax)
        ctx.mode, ctx.ncls = mode, ncls
        ctx.sinks = _sinks(ctx, (w, b), 1)
        ctx.set_materialize_grads(False)
        return frame, clip

    @staticmethod
    def backward(ctx, g_frame, g_clip):
        feat, wpt, frame, amax = ctx.saved_tensors
        B, T, C = feat.shape
        ncls = ctx.ncls
        g_clip = _f32c(g_clip) if g_clip is not None else torch.zeros((B, ncls), dtype=torch.float32, device=feat.device)
        gl = torch.empty((B * T, LDN), dtype=torch.float32, device=feat.device)
        _call("sed_head_pool_bwd", _ptr(g_clip), _ptr(frame), _ptr(amax), B, T, LDN, ncls, ctx.mode, _ptr(gl), _stream())
        if g_frame is not None:                       # d frame / d logit = frame (1 - frame)
            gl.view(B, T, LDN)[:, :, :ncls] += g_frame * frame * (1.0 - frame)
        g_feat = gemm_nt(gl, wpt).view(B, T, C)
        dwp = gemm_tn(feat.view(B * T, C), gl)
        db = col_sums(gl, ncls, out=_dst(ctx.sinks[1], (ncls,), feat.device))
        return g_feat, _put(ctx.sinks[0], dwp[:ncls]), _ret(ctx.sinks[1], db), None


class AttHeadFn(torch.autograd.Function):
    """AttBlock(n_in, n_out, activation, temperature) (models.py:118-149; every model of the reference uses 'sigmoid', 1.0).
    feat (B,T,n_in) -> clip (B,n_out), cla (B,T,n_out), norm_att (B,T,n_out).  The training loss (clip_bce) feeds `clip`; gradients
    arriving through `cla` / `norm_att` (strong-label losses, not used by the reference's main.py) are honoured too, off the
    hot path."""

    @staticmethod
    def forward(ctx, feat, w_att, b_att, w_cla, b_cla, activation="sigmoid", temperature=1.0):
        if activation not in ("linear", "sigmoid") or not temperature > 0:
            raise Exception("Incorrect argument!")
        ctx.act, ctx.temp = (1 if activation == "sigmoid" else 0), float(temperature)
        _chk_dev(feat, w_att)
        feat = _f32c(feat)
        B, T, C = feat.shape
        ncls = w_att.shape[0]
        dev = feat.device
        wp, wpt = _pad_rows([w_att, w_cla], dev)
        logits = gemm_nt(feat.view(B * T, C), wp)
        clip = torch.empty((B, ncls), dtype=torch.float32, device=dev)
        cla = torch.empty((B, T, ncls), dtype=torch.float32, device=dev)
        natt = torch.empty((B, T, ncls), dtype=torch.float32, device=dev)
        asum = torch.empty((B, ncls), dtype=torch.float32, device=dev)
        b_att, b_cla = _f32c(b_att), _f32c(b_cla)
        _call("sed_att_pool_fwd", _ptr(logits), B, T, LDN, ncls, _ptr(b_att), _ptr(b_cla), _ptr(clip), _ptr(cla), _ptr(natt),
              _ptr(asum), ctx.act, ctx.temp, _stream())
        ctx.save_for_backward(feat, wpt, logits, b_att, clip, cla, natt, asum)
        ctx.ncls, ctx.wshape = ncls, w_att.shape
        ctx.sinks = _sinks(ctx, (w_att, b_att, w_cla, b_cla), 1)
        ctx.set_materialize_grads(False)
        return clip, cla, natt

    @staticmethod
    def backward(ctx, g_clip, g_cla, g_natt):
        feat, wpt, logits, b_att, clip, cla, natt, asum = ctx.saved_tensors
        B, T, C = feat.shape
        ncls = ctx.ncls
        g_clip = _f32c(g_clip) if g_clip is not None else torch.zeros((B, ncls), dtype=torch.float32, device=feat.device)
        gl = torch.empty((B * T, LDN), dtype=torch.float32, device=feat.device)
        _call("sed_att_pool_bwd", _ptr(g_clip), _ptr(logits), _ptr(b_att), _ptr(clip), _ptr(cla), _ptr(natt), _ptr(asum), B, T,
              LDN, ncls, _ptr(gl), ctx.act, ctx.temp, _stream())
        gl3 = gl.view(B, T, LDN)
        if g_cla is not None:                         # cla = sigmoid(.) or the identity
            gl3[:, :, ncls:2 * ncls] += (g_cla * cla * (1.0 - cla)) if ctx.act else g_cla
        if g_natt is not None:                        # norm_att = a / sum_t a,  a = exp(clamp(z, -10, 10) / temperature) + 1e-6
            S = asum.view(B, 1, ncls)
            da = (g_natt - (g_natt * natt).sum(dim=1, keepdim=True)) / S
            z = logits.view(B, T, LDN)[:, :, :ncls] + b_att.view(1, 1, ncls)
            gl3[:, :, :ncls] += da * (natt * S - 1e-6) * ((z >= -10.0) & (z <= 10.0)).to(da.dtype) / ctx.temp
        g_feat = gemm_nt(gl, wpt).view(B, T, C)
        dwp = gemm_tn(feat.view(B * T, C), gl)
        dbias = col_sums(gl, 2 * ncls)
        sk = ctx.sinks
        return (g_feat, _put(sk[0], dwp[:ncls].reshape(ctx.wshape)), _put(sk[1], dbias[:ncls]),
                _put(sk[2], dwp[ncls:2 * ncls].reshape(ctx.wshape)), _put(sk[3], dbias[ncls:2 * ncls]), None, None)


class MultiHeadFn(torch.autograd.Function):
    """MultiHead(n_head=8, d_model=512, d_k=d_v=64).forward(x, x, x) of the Transformer heads (models.py:641-665):
    q/k/v projections (MFMA GEMMs), scaled dot-product attention with attention dropout, output projection, dropout,
    ReLU.  x (B,T,512) -> (B,T,512).  `keep_attn` (8*B,T,T) / `keep_fc` (B,T,512) are bool KEEP masks (None = no dropout)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, keep_attn, keep_fc, p_attn, p_fc):
        _chk_dev(x, wq)
        x = _f32c(x)
        B, T, C = x.shape
        M = B * T
        x2 = x.view(M, C)
        xa = amax_of(x2) if gemm_nt_sf16_ok(M, C, C) else None         # one pass serves the three projections
        q, k, v = (linear_nt(x2, w_, _f32c(b_), x_amax=xa) for w_, b_ in ((wq, bq), (wk, bk), (wv, bv)))
        o = torch.empty((M, C), dtype=torch.float32, device=x.device)
        stats = torch.empty((B, 8, T, 4), dtype=torch.float32, device=x.device)
        ka = keep_attn.contiguous() if keep_attn is not None else None
        kf = keep_fc.contiguous() if keep_fc is not None else None
        if ka is not None and (ka.dtype not in (torch.bool, torch.uint8) or tuple(ka.shape) != (8 * B, T, T)):
            raise RuntimeError("attention keep mask must be bool/uint8 of shape (8*B, T, T)")
        if kf is not None and (kf.dtype not in (torch.bool, torch.uint8) or kf.numel() != M * C):
            raise RuntimeError("fc keep mask must be bool/uint8 of shape (B, T, 512)")
        nbits = _lib.lib().sed_mha_mask_words(B, T) if ka is not None else 0
        kbits = torch.empty((nbits,), dtype=torch.int32, device=x.device) if nbits else None      # the mask as bits (MFMA kernels)
        _call("sed_mha_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(ka), float(p_attn), B, T, _ptr(o), _ptr(stats), _ptr(kbits), _stream())
        ctx.kbits = kbits
        y = linear_nt(o, wo, _f32c(bo))
        out = torch.empty_like(y)
        _call("sed_drop_relu_fwd", _ptr(y), _ptr(kf), float(p_fc), M * C, _ptr(out), _stream())
        ctx.save_for_backward(x2, q, k, v, o, stats, out, wq, wk, wv, wo, ka, kf)
        ctx.dims, ctx.p = (B, T, C), (float(p_attn), float(p_fc))
        ctx.xa = xa
        ctx.sinks = _sinks(ctx, (wq, bq, wk, bk, wv, bv, wo, bo), 1)
        return out.view(B, T, C)

    @staticmethod
    def backward(ctx, g):
        x2, q, k, v, o, stats, out, wq, wk, wv, wo, ka, kf = ctx.saved_tensors
        B, T, C = ctx.dims
        M = B * T
        g = _f32c(g).view(M, C)
        gy = torch.empty_like(g)
        _call("sed_drop_relu_bwd", _ptr(g), _ptr(out), _ptr(kf), ctx.p[1], M * C, _ptr(gy), _stream())
        sk = ctx.sinks                       # (wq, bq, wk, bk, wv, bv, wo, bo)
        sf = ctx.xa is not None
        gya = amax_of(gy) if sf else None                  # one pass serves the weight gradient and the input gradient
        dwo = _ret(sk[6], gemm_tn(o, gy, out=_dst(sk[6], (C, C), g.device), gy_amax=gya))
        dbo = _ret(sk[7], col_sums(gy, out=_dst(sk[7], (C,), g.device)))
        go = linear_nt(gy, wo, transposed=True, x_amax=gya)
        gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        _call("sed_mha_bwd", _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(go), _ptr(ka), ctx.p[0], B, T, _ptr(stats), _ptr(gq),
              _ptr(gk), _ptr(gv), _ptr(ctx.kbits), _stream())
        gas = [amax_of(t_) if sf else None for t_ in (gq, gk, gv)]
        gx = linear_nt(gq, wq, transposed=True, x_amax=gas[0])
        for gt, w, ga in ((gk, wk, gas[1]), (gv, wv, gas[2])):
            t = linear_nt(gt, w, transposed=True, x_amax=ga)
            _call("sed_axpy", _ptr(gx), _ptr(t), M * C, _stream())
        grads = []
        for n, gt in enumerate((gq, gk, gv)):
            grads.append(_ret(sk[2 * n], gemm_tn(x2, gt, out=_dst(sk[2 * n], (C, C), g.device), x_amax=ctx.xa, gy_amax=gas[n])))
            grads.append(_ret(sk[2 * n + 1], col_sums(gt, out=_dst(sk[2 * n + 1], (C,), g.device))))
        return (gx.view(B, T, C),) + tuple(grads) + (dwo, dbo, None, None, None, None)


def gemm_nt_pair(x0, x1, w0, w1, b0, b1, y0, y1):
    """Two independent y = x w^T (+b) of one shape in ONE launch (the two GRU directions)."""
    M, K = x0.shape
    N = w0.shape[0]
    _call("sed_gemm_nt_pair", _ptr(x0), _ptr(x1), _ptr(w0), _ptr(w1), _ptr(b0), _ptr(b1), _ptr(y0), _ptr(y1), M, N, K, _stream())


def _gru_ih_operands(w_ih_f, w_ih_b, b_ih_f, b_ih_b):
    """(6H, I) stacked input weights of both directions, their (6H,) biases and the (I, 6H) transpose."""
    w_ih = torch.cat([_f32c(w_ih_f.detach()), _f32c(w_ih_b.detach())], dim=0).contiguous()
    b_ih = torch.cat([_f32c(b_ih_f.detach()), _f32c(b_ih_b.detach())], dim=0).contiguous()
    n, i = w_ih.shape
    return w_ih, b_ih, transpose_b(w_ih.view(1, n, i)).view(i, n)


class GruFn(torch.autograd.Function):
    """nn.GRU(512, 256, num_layers=1, bias=True, batch_first=True, bidirectional=True), h0 = 0
    (models.py:529-530, :565-567).  x (B,T,512) -> (B,T,512) = concat(forward, backward).
    Input projections of both directions = one MFMA GEMM; every recurrence step = one paired GEMM launch + one paired
    gate launch for BOTH directions (forward at time s, reverse at time T-1-s)."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_b, w_hh_b, b_ih_b, b_hh_b, x_amax=None):
        """x_amax (optional): device amax vector of x left by its producer (block 4's pool kernel): the split-f16 input projection
        takes its operand scale from it instead of a pass over x."""
        _chk_dev(x, w_ih_f)
        x = _f32c(x)
        B, T, I = x.shape
        Hd = w_hh_f.shape[1]
        dev = x.device
        out = torch.empty((B, T, 2 * Hd), dtype=torch.float32, device=dev)
        check_device_errors(nonfinite=False)                                           # a give-up of an earlier fused launch
        # stacked / transposed weight operands: device copies, rebuilt only when a parameter changed
        w_ih, b_ih, w_ih_t = _cached("gru_ih", (w_ih_f, w_ih_b, b_ih_f, b_ih_b), lambda: _gru_ih_operands(
            w_ih_f, w_ih_b, b_ih_f, b_ih_b))
        sf_proj = gemm_nt_sf16_ok(B * T, 6 * Hd, I) and gemm_nt_sf16_ok(B * T, I, 6 * Hd)
        if sf_proj and x_amax is None:
            x_amax = amax_of(x)
        ctx.x_amax = x_amax if sf_proj else None
        if sf_proj:          # input projections of both directions on the f16 MFMA pipe (split-f16 operands); packs cached per step
            pk_f, pk_t = _cached("gru_ih_sf16", (w_ih_f, w_ih_b), lambda: (gemm_pack_sf16(w_ih), gemm_pack_sf16(w_ih_t)))
            gi = gemm_nt_sf16(x.view(B * T, I), pk_f, 6 * Hd, b_ih, x_amax=x_amax).view(B, T, 6 * Hd)
            ctx.pk_t = pk_t
        else:
            gi = gemm_nt(x.view(B * T, I), w_ih, b_ih).view(B, T, 6 * Hd)              # (B, T, 6H)
            ctx.pk_t = None
        hs = torch.empty((2, T, B, Hd), dtype=torch.float32, device=dev)
        whh = (_f32c(w_hh_f), _f32c(w_hh_b))
        bhh = (_f32c(b_hh_f), _f32c(b_hh_b))
        ctx.sinks = _sinks(ctx, (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_b, w_hh_b, b_ih_b, b_hh_b), 1)
        ctx.fused = bool(USE_FUSED_GRU and _lib.lib().sed_gru_seq_supported(B, Hd))
        # saved gates r, z, n, gh_n: the fused pair of recurrences keeps them in a private tile layout (rows padded to its row block)
        saves = (torch.empty((_lib.lib().sed_gru_seq_saves_floats(B, T),), dtype=torch.float32, device=dev) if ctx.fused
                 else torch.empty((2, T, B, 4 * Hd), dtype=torch.float32, device=dev))
        if ctx.fused:
            # fused recurrence: ONE persistent launch for all T steps of both directions (csrc/gru.hip); a launch that
            # cannot make progress poisons `out` with NaN and raises the host-mapped flag polled by check_device_errors
            ws = torch.empty((_lib.lib().sed_gru_seq_ws_floats(),), dtype=torch.float32, device=dev)
            with _timed("gru_recurrence_fwd|B%d T%d", (B, T), 2.0 * 2 * B * T * 3 * Hd * Hd):
                _call("sed_gru_seq_fwd", _ptr(gi), _ptr(whh[0]), _ptr(whh[1]), _ptr(bhh[0]), _ptr(bhh[1]), B, T, Hd,
                      _ptr(hs), _ptr(saves), _ptr(out), _ptr(ws), _ptr(_err_flag()), _stream())
            ctx.save_for_backward(x, w_ih_t, whh[0], whh[1], hs, saves)
            return out
        gh0 = torch.stack([bhh[0].view(1, -1).expand(B, -1), bhh[1].view(1, -1).expand(B, -1)]).contiguous()  # h0 = 0
        gh = torch.empty((2, B, 3 * Hd), dtype=torch.float32, device=dev)
        s = _stream()
        for k in range(T):
            tf, tb = k, T - 1 - k
            if k == 0:
                g0, g1, p0, p1 = gh0[0], gh0[1], None, None
            else:
                p0, p1 = hs[0, tf - 1], hs[1, tb + 1]
                gemm_nt_pair(p0, p1, whh[0], whh[1], bhh[0], bhh[1], gh[0], gh[1])
                g0, g1 = gh[0], gh[1]
            _call("sed_gru_gate_fwd", _ptr(gi[:, tf, 0:3 * Hd]), _ptr(gi[:, tb, 3 * Hd:6 * Hd]), T * 6 * Hd, _ptr(g0), _ptr(g1),
                  _ptr(p0), _ptr(p1), B, Hd, _ptr(hs[0, tf]), _ptr(hs[1, tb]), Hd, _ptr(out[:, tf, 0:Hd]),
                  _ptr(out[:, tb, Hd:2 * Hd]), T * 2 * Hd, _ptr(saves[0, tf]), _ptr(saves[1, tb]), s)
        ctx.save_for_backward(x, w_ih_t, whh[0], whh[1], hs, saves)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, w_ih_t, w_hh_f, w_hh_b, hs, saves = ctx.saved_tensors
        g_out = _f32c(g_out)
        B, T, I = x.shape
        Hd = w_hh_f.shape[1]
        dev = x.device
        s = _stream()
        dgi = torch.empty((B, T, 6 * Hd), dtype=torch.float32, device=dev)
        dgh = torch.empty((2, T, B, 3 * Hd), dtype=torch.float32, device=dev)
        wt = _cached("gru_hh_t", (w_hh_f, w_hh_b), lambda: (                          # (H, 3H): dh = dgh x W_hh
            transpose_b(w_hh_f.view(1, 3 * Hd, Hd)).view(Hd, 3 * Hd), transpose_b(w_hh_b.view(1, 3 * Hd, Hd)).view(Hd, 3 * Hd)))
        fused = ctx.fused
        dbp = None
        dgi_amax = None
        if fused:
            ws = torch.empty((_lib.lib().sed_gru_seq_ws_floats(),), dtype=torch.float32, device=dev)
            rb = _lib.lib().sed_gru_seq_row_block()
            nrb = (B + rb - 1) // rb
            dbp = torch.empty((2, nrb, 4 * Hd), dtype=torch.float32, device=dev)     # bias-gradient sums per row block
            dgi_amax = _amax_buf(dev) if ctx.pk_t is not None else None      # amax of dgi, published by the recurrence itself
            with _timed("gru_recurrence_bwd|B%d T%d", (B, T), 2.0 * 2 * B * T * 3 * Hd * Hd):
                _call("sed_gru_seq_bwd", _ptr(g_out), _ptr(wt[0]), _ptr(wt[1]), _ptr(hs), _ptr(saves), B, T, Hd,
                      _ptr(dgi), _ptr(dgh), _ptr(dbp), _ptr(ws), _ptr(_err_flag()), _ptr(dgi_amax), s)
        direct = [torch.empty((2, B, Hd), dtype=torch.float32, device=dev) for _ in range(2)]   # ping-pong
        rec = [torch.empty((2, B, Hd), dtype=torch.float32, device=dev) for _ in range(2)]
        have = False
        for k in (() if fused else range(T - 1, -1, -1)):   # reverse of the forward processing order
            tf, tb = k, T - 1 - k
            cur, prv = k & 1, (k & 1) ^ 1
            p0 = hs[0, tf - 1] if k > 0 else None
            p1 = hs[1, tb + 1] if k > 0 else None
            _call("sed_gru_gate_bwd", _ptr(g_out[:, tf, 0:Hd]), _ptr(g_out[:, tb, Hd:2 * Hd]), T * 2 * Hd,
                  _ptr(direct[prv][0]) if have else None, _ptr(direct[prv][1]) if have else None,
                  _ptr(rec[prv][0]) if have else None, _ptr(rec[prv][1]) if have else None,
                  _ptr(saves[0, tf]), _ptr(saves[1, tb]), _ptr(p0), _ptr(p1), B, Hd,
                  _ptr(dgi[:, tf, 0:3 * Hd]), _ptr(dgi[:, tb, 3 * Hd:6 * Hd]), T * 6 * Hd, _ptr(dgh[0, tf]), _ptr(dgh[1, tb]),
                  _ptr(direct[cur][0]), _ptr(direct[cur][1]), s)
            if k > 0:
                gemm_nt_pair(dgh[0, tf], dgh[1, tb], wt[0], wt[1], None, None, rec[cur][0], rec[cur][1])
            have = True
        # weight gradients of the recurrence: dW_hh = sum_t dgh_t^T h_{prev(t)}; h_prev is a shifted view of hs
        sk = ctx.sinks                       # (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_b, w_hh_b, b_ih_b, b_hh_b)
        if T > 1:
            # |h| <= 1; |dgh| <= |dgi| element-wise (dgh's third gate is dgi's times r in (0, 1)): both operand scales are free
            ha = unit_amax(dev) if dgi_amax is not None else None
            dw_hh_f = gemm_tn(hs[0, 0:T - 1].reshape((T - 1) * B, Hd), dgh[0, 1:T].reshape((T - 1) * B, 3 * Hd),
                              out=_dst(sk[1], (3 * Hd, Hd), dev), x_amax=ha, gy_amax=dgi_amax)
            dw_hh_b = gemm_tn(hs[1, 1:T].reshape((T - 1) * B, Hd), dgh[1, 0:T - 1].reshape((T - 1) * B, 3 * Hd),
                              out=_dst(sk[5], (3 * Hd, Hd), dev), x_amax=ha, gy_amax=dgi_amax)
        else:
            dw_hh_f, dw_hh_b = _dst(sk[1], (3 * Hd, Hd), dev).zero_(), _dst(sk[5], (3 * Hd, Hd), dev).zero_()
        dgi2 = dgi.view(B * T, 6 * Hd)
        if dbp is not None:
            # the fused recurrence summed the bias gradients over (time, rows of a block) as it went: (dr, dz, dn, dn*r) per
            # direction; db_ih = (dr, dz, dn), db_hh = (dr, dz, dn*r) -- no pass over dgi / dgh
            sums = [col_sums(dbp[dd]) for dd in range(2)]                                # (4H,) each, over the row blocks
            db_ih = torch.cat([sums[0][:3 * Hd], sums[1][:3 * Hd]])
            db_hh_f = _put(sk[3], torch.cat([sums[0][:2 * Hd], sums[0][3 * Hd:]]))
            db_hh_b = _put(sk[7], torch.cat([sums[1][:2 * Hd], sums[1][3 * Hd:]]))
        else:
            db_hh_f = _ret(sk[3], col_sums(dgh[0].view(T * B, 3 * Hd), out=_dst(sk[3], (3 * Hd,), dev)))
            db_hh_b = _ret(sk[7], col_sums(dgh[1].view(T * B, 3 * Hd), out=_dst(sk[7], (3 * Hd,), dev)))
            db_ih = col_sums(dgi2)
        if ctx.pk_t is not None:
            gx = gemm_nt_sf16(dgi2, ctx.pk_t, I, x_amax=dgi_amax).view(B, T, I)
        else:
            gx = gemm_nt(dgi2, w_ih_t).view(B, T, I)
        dw_ih = gemm_tn(x.view(B * T, I), dgi2, x_amax=ctx.x_amax if dgi_amax is not None else None, gy_amax=dgi_amax)   # (6H, I)
        return (gx, _put(sk[0], dw_ih[:3 * Hd]), _ret(sk[1], dw_hh_f), _put(sk[2], db_ih[:3 * Hd]), db_hh_f,
                _put(sk[4], dw_ih[3 * Hd:]), _ret(sk[5], dw_hh_b), _put(sk[6], db_ih[3 * Hd:]), db_hh_b, None)


class ClipBceFn(torch.autograd.Function):
    """losses.py:5-12."""

    @staticmethod
    def forward(ctx, p, y):
        _chk_dev(p, y)
        p, y = _f32c(p), _f32c(y)
        loss = torch.empty((1,), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        _call("sed_clip_bce", _ptr(p), _ptr(y), p.numel(), _ptr(loss), _ptr(grad), _stream())
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        g = _f32c(g)
        if g.numel() != 1 or not g.is_cuda:
            return grad * g, None
        out = torch.empty_like(grad)
        _call("sed_scale_by_scalar", _ptr(grad), _ptr(g), grad.numel(), _ptr(out), _stream())
        return out, None


def mixup_rows(x, lam):
    """pytorch_utils.py:80-93 for a (2B, ...) tensor (the targets)."""
    _chk_dev(x, lam)
    x = _f32c(x)
    lam = _f32c(lam)
    B2 = x.shape[0]
    if B2 == 0 or B2 % 2 or lam.numel() != B2:
        raise ValueError("do_mixup needs an even, non-empty batch and one lambda per row (pytorch_utils.py:80-93): "
                         "got %d rows and %d lambdas" % (B2, lam.numel()))
    D = x.numel() // B2
    out = torch.empty((B2 // 2,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _call("sed_mixup_rows", _ptr(x), _ptr(lam), B2, D, _ptr(out), _stream())
    return out


def adam_amsgrad_(p, g, m, v, vmax, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, guard=None, skipped=None,
                  status_ptr=None, rank_flag=None):
    """guard (default: on while the split-f16 kernels are in use): found-non-finite skip -- the update is refused when a
    split-f16 kernel of this step met a NaN / inf operand or the (all-reduced) gradient holds one; see NonFiniteOperand.
    With ops.USE_SF16 = False the step behaves exactly like torch.optim.Adam (NaN gradients make NaN parameters)."""
    _chk_dev(p, g)
    if guard is None:
        guard = USE_SF16
    invalidate_weight_caches()                     # parameters change through raw pointers: `_version` does not see it
    _call("sed_adam_amsgrad", _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vmax), p.numel(), step, lr, beta1, beta2, eps,
          grad_scale, _sf16_err_dev_ptr(p.device) if guard else None, _ptr(skipped) if guard else None,
          _sf16_err_ptr() if guard else None, ctypes.c_void_p(status_ptr) if (guard and status_ptr) else None,
          _ptr(rank_flag) if guard else None, _stream())


def guard_publish(flag_out):
    """flag_out[0] = NaN if this rank's found-non-finite word is set, else 0 (see include/sed_hip.h: sed_guard_publish)."""
    _call("sed_guard_publish", _sf16_err_dev_ptr(flag_out.device), _ptr(flag_out), _stream())
