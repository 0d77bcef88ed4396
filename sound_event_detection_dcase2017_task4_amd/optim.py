"""Fused Adam(amsgrad=True) over ONE flat parameter buffer (reference optimiser: main.py:144-145, :256-258).

The trainable parameters of the model are re-pointed at slices of a single contiguous fp32 buffer and their
`.grad`s at slices of a single contiguous gradient buffer, so that (i) the optimiser step is one HIP kernel
(sed_adam_amsgrad), (ii) the data-parallel exchange is a handful of RCCL all-reduces over contiguous ranges of that
buffer (parallel.GradBuckets) instead of DataParallel's per-step parameter broadcast + reduce_add, and (iii) the
backward kernels write their weight gradients STRAIGHT into the buffer (`direct_grads`): the autograd Functions of
ops.py find the destination through the `_sed_sink` attribute this optimiser puts on every parameter, write there and
hand autograd `None`, so no per-parameter accumulate kernel runs.
Parameters that never receive a gradient (the reference's unused `att_block.bn_att.*`) keep a zero gradient and
therefore never move, which equals torch.optim.Adam skipping `grad is None` parameters.

`direct_grads=True` means "this backward pass DEFINES the gradient" (what zero_grad() + backward() gives); accumulating
several backward passes into one step needs `direct_grads=False` (plain autograd accumulation into the same views).
"""
import collections

import torch

from . import ops, parallel

GUARD_PAD = 4          # floats in front of the flat gradient: [0] = the rank flag of the found-non-finite guard (16-byte alignment kept)
POLL_RING = 16         # host-mapped status words, one per optimiser step in flight (> poll_lag + 1)


class GradSink(object):
    """Where the gradient of one parameter goes: a view into the flat gradient buffer + the readiness callback."""
    __slots__ = ("opt", "index", "view")

    def __init__(self, opt, index, view):
        self.opt, self.index, self.view = opt, index, view

    def expect(self):
        self.opt.buckets.expect(self.index)

    def done(self):
        self.opt.buckets.ready(self.index)


def bucket_cuts(names, offsets, numels, min_bytes=1 << 19):
    """Cut positions (element offsets) for the gradient buckets: parameters are grouped by top-level module
    (`conv_block3.conv1.weight` -> `conv_block3`), groups are walked from the END of the buffer (the order backward
    finishes them) and a cut is placed in front of a group once the running bucket holds >= min_bytes.
    Cnn_9layers_FrameAvg -> [bn0 + block1 + block2 | block3 | block4 + fc]; Gru_FrameAtt adds [gru + att_block]."""
    groups = []                                   # (start offset, elements) per top-level module, in buffer order
    last = None
    for name, off, k in zip(names, offsets, numels):
        top = name.split(".")[0] if name else None
        if top is None or top != last:
            groups.append([off, 0])
            last = top
        groups[-1][1] += k
    cuts, acc = [], 0
    for start, k in reversed(groups):
        acc += 4 * k
        if acc >= min_bytes and start > 0:
            cuts.append(start)
            acc = 0
    # a small leftover at the very front (bn0 alone) stays with the bucket after it
    if cuts and 4 * min(cuts) < min_bytes:
        cuts.remove(min(cuts))
    return sorted(cuts)


def torch_adam_state_to_flat(sd, numels, trainable):
    """A `torch.optim.Adam(model.parameters(), amsgrad=True).state_dict()` -- the 'optimizer' entry of a checkpoint the REFERENCE
    writes (main.py:144-145, :222-230) -- as flat moment vectors over the TRAINABLE parameters in `model.parameters()` order.

    sd: {'state': {index: {'step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'}}, 'param_groups': [{'params': [indices], 'lr',
    'betas', 'eps', 'amsgrad', ...}]}; numels / trainable: element count / requires_grad of EVERY parameter of the model, in
    `model.parameters()` order (the reference hands Adam the frozen STFT / mel tensors too: they occupy indices, never state).
    Returns (step, exp_avg, exp_avg_sq, max_exp_avg_sq, hyper) with zero moments for parameters that never received a gradient
    (torch keeps no state for them: the unused `att_block.bn_att.*`).  Raises ValueError with the reason when the state does not
    fit this model or is not Adam-amsgrad."""
    import torch
    groups = sd.get("param_groups")
    state = sd.get("state")
    if not isinstance(groups, (list, tuple)) or not isinstance(state, dict) or len(groups) != 1:
        raise ValueError("not a torch.optim.Adam state_dict with ONE parameter group (keys: %s)" % sorted(sd.keys()))
    grp = groups[0]
    if not grp.get("amsgrad", False):
        raise ValueError("the checkpoint's optimiser is Adam WITHOUT amsgrad: no max_exp_avg_sq to resume from "
                         "(the reference trains with amsgrad=True, main.py:144-145)")
    if float(grp.get("weight_decay", 0.0)) != 0.0:
        raise ValueError("the checkpoint's optimiser uses weight_decay=%g; FusedAdamAmsgrad implements weight_decay=0 only" % grp["weight_decay"])
    idx = list(grp["params"])
    if len(idx) != len(numels):
        raise ValueError("the checkpoint's optimiser holds %d parameters, this model has %d (model.parameters() order, frozen "
                         "front-end tensors included)" % (len(idx), len(numels)))
    total = sum(k for k, t in zip(numels, trainable) if t)
    out = [torch.zeros((total,), dtype=torch.float32) for _ in range(3)]
    steps = set()
    off = 0
    for pos, (i, k, t) in enumerate(zip(idx, numels, trainable)):
        st = state.get(i)
        if st is not None and not t:
            raise ValueError("the checkpoint holds Adam moments for parameter %d, which is frozen in this model" % pos)
        if t:
            if st is not None:
                for dst, key in zip(out, ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")):
                    if key not in st:
                        raise ValueError("parameter %d of the checkpoint's optimiser has no '%s'" % (pos, key))
                    v = torch.as_tensor(st[key]).detach().to("cpu", torch.float32).reshape(-1)
                    if v.numel() != k:
                        raise ValueError("parameter %d: the checkpoint's %s has %d elements, the model's parameter %d" % (pos, key, v.numel(), k))
                    dst[off:off + k].copy_(v)
                steps.add(int(round(float(torch.as_tensor(st["step"]).item()))))
            off += k
    if len(steps) > 1:
        raise ValueError("the checkpoint's parameters are at different Adam steps %s: one flat step counter cannot resume them" % sorted(steps))
    hyper = {"lr": float(grp["lr"]), "betas": tuple(float(b) for b in grp["betas"]), "eps": float(grp["eps"])}
    return (steps.pop() if steps else 0), out[0], out[1], out[2], hyper


class FusedAdamAmsgrad(object):
    """poll_lag: how the host learns that the Adam kernel refused a step (found-non-finite guard of the split-f16 path).
    None (default for one process): opportunistically -- ops.check_device_errors() raises at whatever call first sees the
    host-mapped flag.  k >= 0 (default 2 when world_size > 1; the train CLI always uses 2): DETERMINISTICALLY -- step() number
    i waits for the event behind step i - k and reads the status word the Adam kernel of that step left in host-mapped memory.
    Every rank of a data-parallel job refuses the same steps (the rank flag rides on the last gradient bucket) and therefore
    raises ops.NonFiniteOperand from the same step() call, with the same `skipped_steps`: the ranks can re-run those batches
    together and the collectives stay balanced.  The host runs at most k steps ahead of the GPU in this mode."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, world_size=1, direct_grads=True, cuts="auto", poll_lag="auto"):
        if hasattr(model, "named_parameters"):
            named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        else:
            named = [("", p) for p in model if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        names = [n for n, _ in named]
        params = [p for _, p in named]
        # every parameter in model.parameters() order, frozen ones included: how torch.optim.Adam numbers them (checkpoint interchange)
        every = list(model.parameters()) if hasattr(model, "parameters") else list(params)
        self._all_numels = [p.numel() for p in every]
        self._all_trainable = [bool(p.requires_grad) for p in every]
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdamAmsgrad: move the model to the GPU first (HIP kernel, no CPU path)")
        self.params = params
        self.lr, self.betas, self.eps = lr, betas, eps
        self.world_size = world_size
        self.direct_grads = bool(direct_grads)
        n = sum(p.numel() for p in params)
        self.flat = torch.empty((n,), dtype=torch.float32, device=dev)
        self._grad_store = torch.zeros((GUARD_PAD + n,), dtype=torch.float32, device=dev)    # [rank flag, pad | gradients]
        self.flat_grad = self._grad_store[GUARD_PAD:]
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.max_exp_avg_sq = torch.zeros_like(self.flat)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                self.offsets.append(off)
                off += k
        numels = [p.numel() for p in params]
        if cuts == "auto":
            cuts = bucket_cuts(names, self.offsets, numels)
        self.buckets = parallel.GradBuckets(self.flat_grad, self.offsets, numels, cuts, store=self._grad_store, pad=GUARD_PAD)
        self.buckets.pre_fire_check = self._assert_joined
        self.buckets.publish_flag = self._publish_flag
        for i, p in enumerate(params):
            p._sed_sink = GradSink(self, i, p.grad) if self.direct_grads else None
        self.step_count = 0
        self.skipped_steps = 0             # optimiser steps the Adam kernel refused (device error word set, see step())
        self._skipped = torch.zeros((1,), dtype=torch.int32, device=dev)    # ... counted here by the kernel itself
        ops._GUARDED.add(self)             # ops.check_device_errors() books them back, whoever happens to poll
        ops.invalidate_weight_caches()     # parameters moved into the flat buffer: operands derived from them are stale
        self.poll_lag = (2 if world_size > 1 else None) if poll_lag == "auto" else poll_lag
        if self.poll_lag is not None and not (0 <= int(self.poll_lag) < POLL_RING - 1):
            raise ValueError("poll_lag must be None or 0 .. %d" % (POLL_RING - 2))
        self._status = torch.zeros((POLL_RING,), dtype=torch.int32).pin_memory()   # cumulative refused steps, written by the Adam kernel
        self._inflight = collections.deque()                                        # (issue number, event behind that step's Adam kernel)
        self._issued = 0

    def _publish_flag(self):
        """Called by the buckets right before the LAST one goes to the backend: this rank's found-non-finite word -> the flag
        element in front of the gradient (NaN / 0), which that all-reduce carries to every rank."""
        if ops.USE_SF16:
            ops.guard_publish(self._grad_store[:1])

    def _assert_joined(self, bucket, indices):
        """A bucket must not go to RCCL while a weight gradient inside it is still running on the side stream and the main
        stream has not waited for it (the collective is ordered behind the MAIN stream only)."""
        pending = ops.pending_sink_indices(self)
        late = pending.intersection(indices)
        if late:
            raise RuntimeError("gradient bucket %d would be all-reduced before the side-stream weight gradients of "
                               "parameters %s were joined" % (bucket, sorted(late)))

    def _view(self, i):
        off = self.offsets[i]
        return self.flat_grad[off:off + self.params[i].numel()].view(self.params[i].shape)

    def zero_grad(self, set_to_none=False):
        ops.drop_pending_wgrads()          # a backward pass that raised may have left side-stream work behind
        self.buckets.new_gradients()
        self._grad_store.zero_()
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):   # re-attach if user code detached the views
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self._view(i)
                if self.direct_grads:
                    p._sed_sink = GradSink(self, i, p.grad)

    def _gather(self):
        """Bring gradients that do NOT already sit in the flat buffer into it.  With direct_grads the backward kernels
        wrote theirs through the sinks and handed autograd None, so after `model.zero_grad()` (set_to_none) such a
        parameter has `p.grad is None` although its slice holds this step's gradient: re-attach the view, never zero it.
        Only slices that received nothing this cycle are cleared."""
        written = self.buckets.written if self.direct_grads else ()
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            sl = self.flat_grad[off:off + p.numel()]
            if p.grad is None:
                if i in written:
                    p.grad = self._view(i)
                else:
                    sl.zero_()
            elif p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                g = p.grad.reshape(-1)
                if i in written:           # a sink write AND an autograd-delivered gradient (parameter used twice)
                    sl.add_(g)
                else:
                    sl.copy_(g)
                p.grad = self._view(i)

    def reduce_gradients(self):
        """Finish the data-parallel exchange of this step's gradients (buckets not yet triggered by the backward pass
        are issued now; the current stream then waits for all of them).  Called by step(); idempotent."""
        self._gather()
        self.buckets.finish()

    @torch.no_grad()
    def step(self):
        """One Adam-amsgrad update.  Found-non-finite guard: the kernel reads the device error word of the split-f16
        convolutions and leaves parameters and moments untouched when a kernel of this step met a NaN / inf operand; the
        host learns about it (no synchronisation: host-mapped flag) at this or a later poll -- here or anywhere else
        ops.check_device_errors() is called -- which raises ops.NonFiniteOperand after taking the refused steps back out
        of `step_count`."""
        self.reduce_gradients()
        self.buckets.begin_step()
        self.step_count += 1
        lagged = self.poll_lag is not None and ops.USE_SF16
        ops.adam_amsgrad_(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq, self.step_count,
                          self.lr, self.betas[0], self.betas[1], self.eps, 1.0 / float(self.world_size), skipped=self._skipped,
                          status_ptr=(self._status.data_ptr() + 4 * (self._issued % POLL_RING)) if lagged else None,
                          rank_flag=self._grad_store if self.world_size > 1 else None)
        ops.restore_bn_if_refused()        # a step refused after its forward pass takes the BatchNorm statistics back too
        if lagged:
            ev = torch.cuda.Event()
            ev.record()
            self._inflight.append((self._issued, ev))
            self._issued += 1
            self.poll(self.poll_lag)
            ops.check_device_errors(nonfinite=False)       # the other run-time flags (fused GRU give-up)
        else:
            ops.check_device_errors()      # NonFiniteOperand: step_count / skipped_steps were already corrected there

    def poll(self, lag=0):
        """Deterministic poll of the found-non-finite guard: look at every optimiser step that is at least `lag` steps behind
        the newest one (waiting for its event) and raise ops.NonFiniteOperand if the Adam kernel refused it -- together with
        every step issued since, which the sticky device flag made it refuse too.  poll(0) drains: call it on ALL ranks
        before anything that must not see a half-reported state (evaluation, checkpoints, the end of training)."""
        while self._inflight and (self._issued - 1 - self._inflight[0][0]) >= lag:
            n, ev = self._inflight.popleft()
            ev.synchronize()
            if int(self._status[n % POLL_RING]) > 0:
                torch.cuda.synchronize()
                k = int(self._skipped.item())              # == self._issued - n on every rank
                self._skipped.zero_()
                ops.clear_nonfinite_flags()
                self._status.zero_()
                self._inflight.clear()
                self.step_count = max(0, self.step_count - k)
                self.skipped_steps += k
                raise ops.NonFiniteOperand(
                    "sound_event_detection_dcase2017_task4_amd: a split-f16 convolution (or a BatchNorm statistic) of optimiser "
                    "step %d met NaN / inf on some rank; the Adam kernel refused that step and the %d issued since (%d in all) "
                    "on EVERY rank -- parameters, moments and BatchNorm running statistics are those from before it.  Re-run "
                    "the last %d batches (the train CLI does so on the fp32 MFMA kernels, ops.USE_SF16 = False)."
                    % (self.step_count + 1, k - 1, k, k), k)

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "max_exp_avg_sq": self.max_exp_avg_sq}

    def torch_state_dict(self):
        """The same state in `torch.optim.Adam(model.parameters(), amsgrad=True).state_dict()` layout -- what the reference writes
        into its checkpoints (main.py:222-230) -- for tools that expect it: per-parameter views of the flat moments, indices in
        model.parameters() order (frozen tensors occupy an index and carry no state)."""
        state, ours = {}, 0
        for i, (k, t) in enumerate(zip(self._all_numels, self._all_trainable)):
            if not t:
                continue
            off, shape = self.offsets[ours], self.params[ours].shape
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + k].view(shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view(shape).clone(),
                        "max_exp_avg_sq": self.max_exp_avg_sq[off:off + k].view(shape).clone()}
            ours += 1
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0.0, "amsgrad": True, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self._all_numels)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts this class's own state_dict() AND a stock `torch.optim.Adam(amsgrad=True)` one (the 'optimizer' entry of a
        checkpoint written by the reference, main.py:222-230: {'state', 'param_groups'}); anything else is refused with the reason
        (ValueError) -- never half-loaded.  Moments and the step counter are taken over; lr / betas / eps stay those this
        optimiser was constructed with (the train CLI's flags decide, as in the reference, which never reloads its optimiser)."""
        if "exp_avg" in sd and "step" in sd:
            step = int(sd["step"])
            moments = [torch.as_tensor(sd[k]) for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")]
            for k, m in zip(("exp_avg", "exp_avg_sq", "max_exp_avg_sq"), moments):
                if m.numel() != self.flat.numel():
                    raise ValueError("optimiser state '%s' has %d elements, this model's flat buffer %d" % (k, m.numel(), self.flat.numel()))
        elif "state" in sd and "param_groups" in sd:
            step, m1, m2, m3, hyper = torch_adam_state_to_flat(sd, self._all_numels, self._all_trainable)
            moments = [m1, m2, m3]
        else:
            raise ValueError("unknown optimiser state layout (keys: %s): expected FusedAdamAmsgrad.state_dict() or a "
                             "torch.optim.Adam(amsgrad=True).state_dict()" % sorted(sd.keys()))
        self.step_count = step
        for k, m in zip(("exp_avg", "exp_avg_sq", "max_exp_avg_sq"), moments):
            getattr(self, k).copy_(m.reshape(-1))
        ops.invalidate_weight_caches()
