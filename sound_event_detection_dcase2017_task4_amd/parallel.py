"""Data parallelism for the hot path: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl").

Replaces the reference's single-process `torch.nn.DataParallel` (main.py:138, :337), which per step broadcasts all
parameters + buffers (23-28 MB incl. the 4.3 MB frozen DFT/mel tensors), scatters inputs, gathers outputs and
reduce-adds gradients onto GPU 0 from GIL-bound threads.  Here clips shard across ranks (each rank owns an even number
of waveforms, so mixup pairs (2i, 2i+1) never straddle ranks), BatchNorm statistics stay rank-local (which IS
DataParallel's semantics: per-replica statistics), and the only exchange is the all-reduce of the flat fp32 gradient
buffer of optim.FusedAdamAmsgrad (18.8-23.6 MB), issued in a few contiguous BUCKETS ordered head -> block 1: a bucket
is handed to RCCL (its own stream; the collective waits for the gradient kernels enqueued so far and runs beside the
rest of the backward pass) as soon as the last gradient inside it has been enqueued.  The 1/world scaling is fused
into the Adam kernel.  Works with any backend (gloo on CPU tensors is used by the CPU tests of this logic).
"""
import datetime
import os
import socket
import sys

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_minutes=60):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (rank, world, local_rank).
    The process-group timeout is long on purpose: rank 0 evaluates for minutes every 1000 iterations while the other
    ranks wait at a barrier."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SED_SHARE_GPU") == "1":
        # test / debug switch: all ranks use GPU 0 and talk over gloo, so that the N-rank code paths (CLI, bench) can run on
        # a 1-GPU box -- RCCL refuses two ranks on one device.  Never set in production.
        local_rank, backend = 0, "gloo"
    # SED_FORCE_DIST=1 (tests): build the process group and issue every collective even with ONE rank, so that the RCCL code path
    # (ProcessGroupNCCL with device_id, async all-reduce handles against our streams, broadcasts, barrier, shutdown) executes on
    # a 1-GPU box.  (A one-rank all-reduce moves no bytes: it proves the plumbing, not the bandwidth.)
    if (world > 1 or os.environ.get("SED_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"timeout": datetime.timedelta(minutes=timeout_minutes)}
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                raise RuntimeError("rank %d wants GPU %d but this node exposes %d GPU(s): one process per GPU"
                                   % (rank, local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_torchrun(n_gpus, script, argv):
    """`python script --gpus N` started WITHOUT a launcher: re-execute it as N ranks (one per GPU) of one node under
    torch.distributed.run, replacing this process.  Raises if the node has fewer than N GPUs -- it never degrades to
    fewer ranks."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n_gpus:
        raise SystemExit("%s: --gpus %d needs %d GPUs on this node, found %d (one process per GPU over RCCL; refusing "
                         "to run with fewer ranks)" % (os.path.basename(script), n_gpus, n_gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def shutdown():
    """Orderly end of a multi-rank job: every rank arrives, then the process group (its RCCL communicators and streams)
    is torn down before the interpreter exits, so that no rank is left inside a collective or prints a leaked-group
    warning.  No-op for a single process."""
    if dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def collectives_on():
    """Do the exchange steps run?  With more than one rank -- or with ONE rank when SED_FORCE_DIST=1 built a process group (tests)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SED_FORCE_DIST") == "1")


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if collectives_on():
        dist.barrier()


def broadcast_flat(flat, src=0):
    """Rank `src`'s flat parameter buffer becomes everyone's (one-time, at start)."""
    if collectives_on():
        dist.broadcast(flat, src=src)
    from . import ops                      # in-place write through the flat buffer: neither `_version` of the parameter
    ops.invalidate_weight_caches()         # views nor the optimiser generation saw it (padded head / stacked GRU operands)


def broadcast_buffers(module, src=0):
    if collectives_on():
        for b in module.buffers():
            dist.broadcast(b, src=src)


def broadcast_rng_state(src=0):
    """Every rank continues with rank `src`'s global torch CPU generator state: the SpecAugment positions of a global
    batch are drawn from it on every rank and must agree."""
    if collectives_on():
        state = torch.get_rng_state()
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = state.to(dev)
        dist.broadcast(t, src=src)
        torch.set_rng_state(t.cpu())


def allreduce_flat_grad(flat_grad, buckets=1, async_op=False):
    """Sum the flat gradient buffer over ranks (the mean's 1/world is applied inside the optimiser kernel).
    `buckets` > 1 splits the buffer into contiguous chunks.  (The overlapped, readiness-driven form is GradBuckets.)"""
    if world_size() == 1:
        return []
    n = flat_grad.numel()
    handles = []
    step = (n + buckets - 1) // buckets
    for i in range(0, n, step):
        h = dist.all_reduce(flat_grad[i:i + step], op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


class GradBuckets(object):
    """Readiness-driven bucketed all-reduce of one flat gradient buffer, overlapped with the backward pass.

    The buffer is cut at parameter boundaries into `len(cuts) + 1` contiguous buckets.  Backward produces gradients
    from the END of the buffer (head, GRU, block 4 -- 75 % of the bytes) towards its start (block 1, bn0); every
    gradient writer calls `ready(i)` for parameter i after enqueueing its kernels, and when the last expected parameter
    of a bucket is ready the bucket's all-reduce is issued with async_op=True: ProcessGroupNCCL runs it on its own
    stream behind an event on the current one, i.e. beside the remaining backward kernels.  `finish()` (before the
    optimiser step) issues whatever was not triggered and makes the current stream wait for all of them.

    Which parameters take part is registered per step by the forward pass (`expect`), so parameters that receive no
    gradient (the reference's unused `att_block.bn_att.*`, models.py:129) never block a bucket."""

    def __init__(self, flat_grad, offsets, numels, cuts, store=None, pad=0):
        """store / pad: `flat_grad` is store[pad:]; store[0] is the rank flag of the found-non-finite guard (optim.py).  It is
        published (publish_flag) right before the LAST bucket of a step goes out and rides on that all-reduce when the
        bucket is the one at the front of the buffer (bucket 0: block 1 / bn0 -- the last to finish in every model), else on
        a 16-byte all-reduce of its own."""
        self.flat_grad = flat_grad
        self.store, self.pad = store, int(pad)
        self.publish_flag = None
        self.offsets, self.numels = list(offsets), list(numels)
        n = flat_grad.numel()
        edges = [0] + sorted(set(int(c) for c in cuts if 0 < int(c) < n)) + [n]
        self.ranges = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
        self.bucket_of = []
        for off in self.offsets:
            self.bucket_of.append(next(b for b, (lo, hi) in enumerate(self.ranges) if lo <= off < hi))
        self.issue_order = []          # bucket indices in the order they were handed to the backend (for tests / logs)
        self.pre_fire_check = None     # callable(bucket index, param indices) run before a bucket goes to the backend
        self.wait_events = None        # bench: list that receives (start, end) CUDA event pairs around the waits of finish()
        # deferred: ready() never fires a bucket, finish() issues them all behind the backward pass (graph.GraphedTrainStep
        # sets it while capturing; SED_ALLREDUCE_OVERLAP=0 selects it for a whole run: the exchange then starts after the last
        # MFMA kernel of backward instead of beside it -- <= 0.3 ms of exposed all-reduce per step)
        self.deferred = os.environ.get("SED_ALLREDUCE_OVERLAP", "1") == "0"
        self.begin_step()
        self.last_issue_order = []

    def begin_step(self):
        self.last_issue_order = list(self.issue_order)
        self.pending = [set() for _ in self.ranges]
        self.fired = [False] * len(self.ranges)
        self.written = set()
        self.rewritten = False
        self.handles = []
        self.issue_order = []

    def new_gradients(self):
        """zero_grad(): the gradients about to be produced replace whatever an earlier backward pass of this cycle left
        (its all-reduces, if any, are drained first).  Expectations registered by the forward pass are kept."""
        for h in self.handles:
            h.wait()
        self.handles = []
        self.fired = [False] * len(self.ranges)
        self.written = set()
        self.rewritten = False
        self.issue_order = []

    def expect(self, i):
        self.pending[self.bucket_of[i]].add(i)

    def ready(self, i):
        if i in self.written:
            self.rewritten = True      # a second backward pass overwrote this gradient before the optimiser step
        self.written.add(i)
        b = self.bucket_of[i]
        pend = self.pending[b]
        if i in pend:
            pend.discard(i)
            if not pend and not self.fired[b] and not self.deferred:
                self._fire(b)

    def _fire(self, b):
        if self.pre_fire_check is not None:
            # ordering contract: every gradient of this bucket is complete IN MAIN-STREAM ORDER before the collective is
            # enqueued behind that stream (side-stream weight gradients must have been joined: ops.join_side_stream)
            self.pre_fire_check(b, [i for i, bb in enumerate(self.bucket_of) if bb == b])
        self.fired[b] = True
        self.issue_order.append(b)
        if collectives_on():
            lo, hi = self.ranges[b]
            last = all(self.fired)
            if last and self.publish_flag is not None and self.store is not None:
                self.publish_flag()                      # this rank's found-non-finite word -> store[0]
                if lo == 0:                              # rides on this bucket
                    self.handles.append(dist.all_reduce(self.store[0:self.pad + hi], op=dist.ReduceOp.SUM, async_op=True))
                    return
                self.handles.append(dist.all_reduce(self.store[0:self.pad], op=dist.ReduceOp.SUM, async_op=True))
            self.handles.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        if self.rewritten:
            for h in self.handles:         # all-reduces already in flight on this buffer finish before the state is dropped
                h.wait()
            self.begin_step()
            raise RuntimeError("backward() ran more than once between two optimiser steps: with direct_grads=True a "
                               "backward pass DEFINES the gradients (it overwrites, and buckets may already have been "
                               "all-reduced).  Use FusedAdamAmsgrad(..., direct_grads=False) to accumulate.")
        for b in reversed(range(len(self.ranges))):
            if not self.fired[b]:
                self._fire(b)
        timed = self.wait_events is not None and self.handles and self.flat_grad.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self.handles:
            h.wait()                   # stream-level wait for nccl (no host block), completion for gloo
        if timed:
            ev[1].record()             # time the compute stream sits behind the collectives = the exposed all-reduce
            self.wait_events.append(ev)
        self.handles = []


def shard_range(total, rank, world):
    """Contiguous, even-sized shard [lo, hi) of `total` units for `rank` (units = mixup pairs or clips)."""
    per = total // world
    return rank * per, (rank + 1) * per


def shard_rows(n_rows, rank, world, pair=True):
    """Rows [lo, hi) of a global batch of `n_rows` waveforms owned by `rank`: equal contiguous slices; with
    `pair` the slice length must be even so that mixup pairs (2i, 2i+1) stay on one rank (pytorch_utils.py:90-91)."""
    if n_rows % world:
        raise ValueError("global batch of %d waveforms does not split evenly over %d ranks" % (n_rows, world))
    per = n_rows // world
    if pair and per % 2:
        raise ValueError("per-rank slice of %d waveforms is odd: mixup pairs would straddle ranks" % per)
    return rank * per, (rank + 1) * per
