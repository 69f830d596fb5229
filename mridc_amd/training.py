"""Data-parallel training step of the CIRIM on the HIP path (one process per GPU, every rank its own slices, ONE
all-reduce(sum) of the flat gradient per step over RCCL/xGMI -- 1.70 MB for CIRIM-8 -- then Adam on the flat buffers).

The reference reaches the same result through pytorch-lightning's DDP (`strategy ddp`, base_cirim_train.yaml) + torch.optim.Adam;
here the tape is torch autograd over the Functions of `mridc_amd.autograd` (all arithmetic in libmridc_amd.so), the gradient
exchange is one `torch.distributed.all_reduce` on a single contiguous buffer, and the optimizer is `mrx_adam_step`."""
import math

import torch

from mridc_amd import _lib
from mridc_amd import autograd as ag


def cirim_l1_loss(cascades_etas, target, time_steps, num_cascades):
    """`CIRIM.process_loss` with accumulate_estimates and the l1 loss (cirim.py:199-247).  The reference multiplies every per-step
    loss by the WHOLE logspace(-1, 0, time_steps) vector and sums (its weighting quirk), so every step weighs sum(logspace) /
    time_steps; cascades are averaged."""
    tgt = target.abs() if target.is_complex() else target
    tgt = (tgt / tgt.abs().max()).abs().float().contiguous()          # target side: data preparation, no gradient
    w = float(torch.logspace(-1, 0, steps=time_steps).sum()) / time_steps / num_cascades
    total = None
    for cascade in cascades_etas:
        for pred in cascade:
            p = torch.view_as_real(pred) if pred.is_complex() else pred
            term = ag.AbsL1Loss.apply(p.contiguous(), tgt)
            total = term if total is None else total + term
    return total * w


class FlatParameters:
    """All trainable parameters of a module as views into ONE contiguous fp32 buffer, and their gradients as views into another:
    the gradient exchange is a single collective and the optimizer a single launch."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p.data)
            p.grad = self.grad[o:o + k].view_as(p.data)
            o += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()


def allreduce_gradients(flat_grad):
    """Sum the flat gradient over the ranks (no-op without a process group).  Returns the world size to divide by."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    return 1


class AdamFlat:
    """torch.optim.Adam (weight_decay 0, amsgrad off) on a FlatParameters buffer: one mrx_adam_step launch."""

    def __init__(self, flat: FlatParameters, lr=1e-3, betas=(0.9, 0.98), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.steps = 0

    def step(self, grad_scale=1.0):
        self.steps += 1
        f = self.flat
        _lib.check(_lib.lib().mrx_adam_step(_lib.ptr(f.flat), _lib.ptr(f.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), f.numel,
                                            self.lr, self.betas[0], self.betas[1], self.eps, self.steps, float(grad_scale),
                                            _lib.stream_ptr()), "mrx_adam_step")
        # the kernel wrote the parameters behind torch's back: bump their version counters, which is what the packed-weight caches
        # (Winograd / 1x1 / fused-layer packs) and autograd's saved-tensor checks key on
        for p in f.params:
            torch.autograd.graph.increment_version(p)


def inverse_sqrt_lr(step, max_steps, base_lr, warmup_ratio=0.1, min_lr=0.0, warmup_steps=None):
    """The reference's InverseSquareRootAnnealing under its WarmupPolicy (core/optim/lr_scheduler.py:68-88,664-671; configured by
    base_cirim_train.yaml:168-172 with warmup_ratio 0.1, min_lr 0): `step` is the scheduler's last_epoch (0 for the first optimizer step).
      step <= warmup_steps (and warmup_steps > 0):  base_lr * (step + 1) / (warmup_steps + 1)
      step  > max_steps:                            min_lr
      otherwise:                                    base_lr / sqrt((step + 1) / (warmup_steps + 1))"""
    if warmup_steps is None:
        warmup_steps = int(warmup_ratio * max_steps)
    if 0 < warmup_steps >= step:
        return base_lr * (step + 1) / (warmup_steps + 1)
    if step > max_steps:
        return min_lr
    return base_lr / math.sqrt((step + 1) / (warmup_steps + 1))


def training_step(model, flat, optimizer, batch, time_steps=None, schedule=None):
    """One data-parallel step: forward (recorded), l1 loss, backward through the HIP kernels, ONE all-reduce of the flat gradient,
    Adam.  `batch`: dict with y, sensitivity_maps, mask, target.  `schedule`: optional dict(max_steps, base_lr, warmup_ratio | warmup_steps,
    min_lr) -- the learning rate of this step is then the reference's InverseSquareRootAnnealing value (inverse_sqrt_lr at the number of
    optimizer steps taken so far).  Returns the loss (0-dim device tensor)."""
    model.train()
    if schedule is not None:
        optimizer.lr = inverse_sqrt_lr(optimizer.steps, schedule["max_steps"], schedule["base_lr"], schedule.get("warmup_ratio", 0.1),
                                       schedule.get("min_lr", 0.0), schedule.get("warmup_steps"))
    flat.zero_grad()
    etas = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
    loss = cirim_l1_loss(etas, batch["target"], model.time_steps, len(model.cirim))
    loss.backward()
    world = allreduce_gradients(flat.grad)
    optimizer.step(grad_scale=1.0 / world)
    return loss.detach()
