"""Forward + loss + backward of one fixed-shape training step as ONE HIP-graph launch (train loop body, reference
pytorch/main.py:233-258).

At the metric's own batch size (32 clips, reference README) a step is ~140 kernels of 5-400 us: a kernel trace shows
~45 idle gaps of ~6 us between them per step (0.45 ms of 9.7 ms, profiles/r03/step_digest_b32_main_stream_only.txt) and
the host spends most of the step enqueueing.  Captured once, the same kernels replay from one hipGraphLaunch with the
dependencies resolved on the device.  What stays OUTSIDE the graph, by design:
  * the inputs -- waveforms, targets, mixup lambdas and the SpecAugment stripe positions are copied into static device
    buffers before every launch (the stripes are still drawn from the global torch CPU generator in the package's order,
    utils/augmentation.py, so a seeded run is unchanged);
  * the optimiser step -- the Adam kernel takes the step number as an argument, and the gradient buckets' all-reduces are
    issued by FusedAdamAmsgrad.step() behind the graph (with several ranks the exchange is therefore NOT overlapped with
    the backward pass in this mode: meant for the small-batch, launch-bound regime);
  * error polling (ops.check_device_errors): host-mapped words, as in eager mode.
The first `eager_steps` calls run eagerly (they are real training steps: lazily built tables, weight packs, workspaces
and the allocator's pools settle), the next call captures and replays.  A change of input shape / dtype, of the model's
training flag or of ops.USE_SF16 drops the graph and starts over.

    step = GraphedTrainStep(model, optimizer, loss_func, mixup=True)
    loss = step(wave, target, lam)          # device tensor, valid until the next call
"""
import torch

from . import ops, parallel
from .pytorch.pytorch_utils import do_mixup

HOP_SIZE, MEL_BINS = 320, 64        # utils/config.py constants the kernels are specialised for (pytorch/models.py checks them)


class GraphCaptureError(RuntimeError):
    """The HIP-graph CAPTURE of the step was refused -- on this rank or on another one of the job.  Nothing of the step the
    caller is at has run (a capture only records), so the caller may run that batch eagerly instead.  Every other exception
    out of GraphedTrainStep.__call__ (eager warm-up steps, the replay, optimizer.step() with its all-reduces and polls) comes
    from a step that HAS been partly or fully applied and must propagate: re-running that batch would update Adam and the
    BatchNorm statistics twice."""


class GraphedTrainStep(object):
    def __init__(self, model, optimizer, loss_func, mixup=True, eager_steps=3, enabled=True):
        self.model, self.opt, self.loss_func = model, optimizer, loss_func
        self.mixup = bool(mixup)
        self.eager_steps = int(eager_steps)
        self.enabled = bool(enabled)
        self.graph = None
        self.loss = None
        self._written = set()
        self._bn_last = None
        self.calls = 0
        self.replays = 0
        self._key = None
        self.wave = self.target = self.lam = self.stripes = None

    # -- static inputs ------------------------------------------------------------------------------------------------
    def _bind(self, wave, target):
        key = (tuple(wave.shape), wave.dtype, tuple(target.shape), target.dtype, wave.device, bool(ops.USE_SF16),
               self.model.training)
        if key != self._key:
            self.reset()
            self._key = key
            self.wave = torch.empty_like(wave, device=wave.device)
            self.target = torch.empty(target.shape, dtype=torch.float32, device=wave.device)
            self.lam = torch.ones((wave.shape[0],), dtype=torch.float32, device=wave.device) if self.mixup else None
            self.stripes = torch.zeros((wave.shape[0], 8), dtype=torch.int32, device=wave.device)
            self.calls = 0

    def reset(self):
        """Forget the captured graph (its private memory pool is released with it)."""
        self.graph = None
        self.loss = None

    def _small(self, dst, src, dtype):
        if torch.is_tensor(src) and src.is_cuda:
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        else:
            dst.copy_(ops.upload_small(src, dst.device, dtype).reshape(dst.shape), non_blocking=True)

    # -- the captured region ------------------------------------------------------------------------------------------
    def _body(self):
        out = self.model(self.wave, self.lam, specaug_stripes=self.stripes)
        tgt = do_mixup(self.target, self.lam) if self.mixup else self.target
        loss = self.loss_func(out, {"target": tgt})
        self.opt.zero_grad()
        loss.backward()
        return loss

    def _capture(self):
        if not self.model.training:
            raise GraphCaptureError("GraphedTrainStep: the model is in eval mode")
        buckets = self.opt.buckets
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        prev, buckets.deferred = getattr(buckets, "deferred", False), True   # no collective inside the graph
        bn_before = ops._BN_LAST               # BatchNorm groups tracked before the capture (normally none: a step has just ended)
        try:
            # thread_local: only THIS thread's actions can invalidate the recording.  Other threads of the process make HIP calls
            # of their own all the time -- the NCCL watchdog polls its events, the input pipeline's producer thread synchronises
            # its copy events and allocates page-locked buffers -- and under the default "global" mode any such call during the
            # few hundred ms of a capture refuses it (seen once in ~10 runs with a process group alive: round 6)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                ops.reset_amax_pool()           # the zero fill of the amax rows used below becomes part of the graph
                loss = self._body()
        except GraphCaptureError:
            raise
        except Exception as err:                # whatever refused the recording: nothing of this step has run
            ops.drop_pending_wgrads()
            raise GraphCaptureError("HIP graph capture of the training step failed: %r" % (err,)) from err
        finally:
            buckets.deferred = prev
            ops.reset_amax_pool()               # ... and eager code never gets rows of the graph's pool
            # the forward pass that was only RECORDED tracked its BatchNorm entries (ops._BN_LAST) although nothing ran: their
            # `cand` tensors hold no statistics yet.  They belong to the replays (below), never to the eager bookkeeping -- a
            # step refused later must not 'restore' running statistics from them
            captured_bn = (ops._BN_LAST or [])[len(bn_before or []):]
            ops._BN_LAST = bn_before
        if ops.pending_sink_indices():
            ops.drop_pending_wgrads()
            raise GraphCaptureError("GraphedTrainStep: side-stream weight gradients were left un-joined by the captured backward pass")
        self.graph, self.loss = g, loss
        # which gradients the captured backward pass writes through the sinks: a replay runs no Python, so the buckets'
        # `written` set must be restored by hand before every optimiser step -- otherwise FusedAdamAmsgrad._gather() would
        # take a `p.grad is None` parameter (model.zero_grad(set_to_none=True) between replays) for one that received
        # nothing and ZERO the slice the graph has just filled
        self._written = set(buckets.written)
        # ... and likewise the BatchNorm entries of the captured forward pass (ops.restore_bn_if_refused() after every replay)
        self._bn_last = captured_bn or None

    def _capture_together(self):
        """Capture, and agree on the outcome with the other ranks: a rank that fell back to the eager loop alone would fire its
        gradient buckets from inside backward while its peers (still graphed) issue theirs behind the graph -- the collective
        sequences would diverge.  One MAX all-reduce of a 'refused' bit (capture time only, never per step): if any rank's
        capture was refused, EVERY rank drops its graph and raises GraphCaptureError from this very call."""
        mine = None
        try:
            self._capture()
        except GraphCaptureError as err:
            mine = err
        if parallel.collectives_on():
            dev = self.wave.device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
            bit = torch.tensor([1.0 if mine is not None else 0.0], dtype=torch.float32, device=dev)
            torch.distributed.all_reduce(bit, op=torch.distributed.ReduceOp.MAX)
            if mine is None and float(bit.item()) > 0:
                mine = GraphCaptureError("the HIP graph capture was refused on another rank: every rank falls back together")
        if mine is not None:
            self.reset()
            self.calls -= 1                     # the caller runs this batch eagerly: not one of this object's steps
            raise mine

    def __call__(self, wave, target, lam=None, stripes=None):
        if self.mixup and lam is None:
            raise ValueError("GraphedTrainStep(mixup=True) needs the mixup lambdas of this batch")
        self._bind(wave, target)
        self.wave.copy_(wave, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        if self.mixup:
            self._small(self.lam, lam, torch.float32)
        if stripes is None:
            frames = wave.shape[1] // HOP_SIZE + 1
            stripes = self.model.spec_augmenter.draw(wave.shape[0], frames, MEL_BINS)
        self._small(self.stripes, stripes, torch.int32)
        self.calls += 1
        if not self.enabled or self.calls <= self.eager_steps:
            loss = self._body()
        else:
            if self.graph is None:
                self._capture_together()        # records only: the replay below is this call's step
            self.opt.buckets.new_gradients()
            self.graph.replay()
            self.replays += 1
            self.opt.buckets.written |= self._written
            ops._BN_LAST = self._bn_last
            loss = self.loss
        self.opt.step()
        return loss
