"""Fused Adam(amsgrad=True) over ONE flat parameter buffer (reference optimiser: main.py:144-145, :256-258).

The trainable parameters of the model are re-pointed at slices of a single contiguous fp32 buffer and their
`.grad`s at slices of a single contiguous gradient buffer, so that (i) the optimiser step is one HIP kernel
(sed_adam_amsgrad) and (ii) the data-parallel exchange is ONE RCCL all-reduce of that buffer
(parallel.allreduce_gradients) instead of DataParallel's per-step parameter broadcast + reduce_add.
Parameters that never receive a gradient (the reference's unused `att_block.bn_att.*`) keep a zero gradient and
therefore never move, which equals torch.optim.Adam skipping `grad is None` parameters.
"""
import torch

from . import ops


class FusedAdamAmsgrad(object):
    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, world_size=1):
        params = [p for p in (model.parameters() if hasattr(model, "parameters") else model) if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdamAmsgrad: move the model to the GPU first (HIP kernel, no CPU path)")
        self.params = params
        self.lr, self.betas, self.eps = lr, betas, eps
        self.world_size = world_size
        n = sum(p.numel() for p in params)
        self.flat = torch.empty((n,), dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros((n,), dtype=torch.float32, device=dev)
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
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):          # re-attach if user code detached the views
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def _gather(self):
        for p, off in zip(self.params, self.offsets):
            if p.grad is None:
                self.flat_grad[off:off + p.numel()].zero_()
            elif p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                self.flat_grad[off:off + p.numel()].copy_(p.grad.reshape(-1))

    @torch.no_grad()
    def step(self):
        self._gather()
        self.step_count += 1
        ops.adam_amsgrad_(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq, self.step_count,
                          self.lr, self.betas[0], self.betas[1], self.eps, 1.0 / float(self.world_size))

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "max_exp_avg_sq": self.max_exp_avg_sq}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            getattr(self, k).copy_(sd[k])
