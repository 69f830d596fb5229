"""Data-parallel training step of the CIRIM on the HIP path (one process per GPU, every rank its own slices, ONE
all-reduce(sum) of the flat gradient per step over RCCL/xGMI -- 1.70 MB for CIRIM-8 -- then Adam on the flat buffers).

The reference reaches the same result through pytorch-lightning's DDP (`strategy ddp`, base_cirim_train.yaml) + torch.optim.Adam;
here the tape is torch autograd over the Functions of `mridc_amd.autograd` (all arithmetic in libmridc_amd.so), the gradient
exchange is one `torch.distributed.all_reduce` on a single contiguous buffer, and the optimizer is `mrx_adam_step`."""
import math
import os

import torch

from mridc_amd import _lib, ops
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
        named = [(n_, p) for n_, p in module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
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
        # contiguous [start, stop) of every cascade's parameters (`cirim.{i}.*` are registered cascade by cascade) and of the others
        self.cascade_slices, self.rest_slices = [], []
        o, ok = 0, True
        for name, p in named:
            parts = name.split(".")
            idx = int(parts[1]) if len(parts) > 2 and parts[0] == "cirim" and parts[1].isdigit() else None
            if idx is None:
                self.rest_slices.append((o, o + p.numel()))
            elif idx == len(self.cascade_slices) - 1:
                self.cascade_slices[-1][1] = o + p.numel()
            elif idx == len(self.cascade_slices):
                self.cascade_slices.append([o, o + p.numel()])
            else:
                ok = False
            o += p.numel()
        self.cascade_slices = [tuple(v) for v in self.cascade_slices] if ok else []

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


# bf16 mode: True = convolution results and their gradients are STORED in bf16 too (pair tensors, the rounding points of torch.autocast:
# _cascade_forward_backward_tl); False = bf16 operands with fp32 storage (the round-2 kernels; a test hook and the fallback for other layer shapes)
BF16_STORAGE = True
# The weight gradients of a time-step hang off the backward chain (cell -> data gradient -> cell -> ... -> adjoint of the likelihood gradient) as side
# branches: they run on a second HIP stream next to it (every kernel of the chain leaves CUs idle: one or two workgroups per CU waiting for their tiles).
# Each stream keeps its own fixed order, so the gradients stay bit-reproducible.
TL_SIDE_STREAM = os.environ.get("MRIDC_AMD_TL_SIDE_STREAM", "1") != "0"      # ("0": one stream -- each kernel's own duration in a profile)
TL_SIDE_IN_CAPTURE = True
_SIDE = {}


def _side_stream(device):
    st = _SIDE.get(str(device))
    if st is None:
        st = _SIDE[str(device)] = torch.cuda.Stream(device=device)
    return st


# ---- the explicit tape ------------------------------------------------------------------------------------------------------------------
# The CIRIM recurrence is a fixed sequence (rim_block.py:217-249), so its backward pass is written out instead of recorded: forward saves the
# activations of one cascade, the backward walks the time-steps in reverse calling the kernels directly.  What that buys over the autograd
# tape (mridc_amd/autograd.py, kept for masks / layers this path does not cover):
#   * no torch kernel anywhere in the step: gradient accumulation happens INSIDE the kernels (weight gradients with accumulate = 1 into the
#     flat gradient buffer, bias / hh sums added by mrx_relu_bwd_acc), the two gradient paths into a hidden state are the two addends of
#     mrx_relu_bwd_acc, permutes and channel splits ride in three small glue kernels;
#   * cascades are detached from each other (cirim.py:146-165 hands `pred[-1].detach()` on), so each cascade's backward runs right after
#     its forward: activations of ONE cascade are alive at a time (~2 GB instead of ~16 GB at 640 x 372), and its slice of the flat
#     gradient is final at that point -- the all-reduce of cascade c overlaps the forward/backward of cascade c + 1.
def _tape_supported(model, batch):
    from mridc_amd.collections.reconstruction.models.rim import rnn_cells
    if not (model.no_dc and model.coil_dim == 1 and ops.mask_is_row_invariant(batch["mask"])):
        return False
    tgt = batch["target"]
    if tuple(tgt.shape[-2:]) != tuple(batch["y"].shape[2:4]):
        return False                                   # a cropped target needs the crop's adjoint: left to the autograd path
    for blk in model.cirim:
        if len(blk.layers) < 1 or blk.final_layer[0] is None:
            return False
        for st in blk.layers:
            c, r = st.convs, st.rnn
            if not (isinstance(r, rnn_cells.IndRNNCell) and r.kernel_size == 1 and c is not None and c.act == ops.ACT_RELU):
                return False
    return True


def _grad_of(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _relu_bwd_acc(dy, dy2, y, h_prev, hh, acc_bias, acc_hh):
    B, C, H, W = [int(v) for v in y.shape]
    L = _lib.lib()
    dpre = torch.empty_like(y)
    dhp = torch.empty_like(y) if h_prev is not None else None
    work = torch.empty(int(L.mrx_relu_bwd_work_floats(C)), dtype=torch.float32, device=y.device)
    hhc = None if h_prev is None else hh.detach().reshape(-1)
    _lib.check(L.mrx_relu_bwd_acc(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(y), _lib.ptr(h_prev), _lib.ptr(hhc), _lib.ptr(dpre), _lib.ptr(dhp),
                                  _lib.ptr(acc_bias), _lib.ptr(acc_hh if h_prev is not None else None), _lib.ptr(work), B, C, H * W,
                                  _lib.stream_ptr()), "mrx_relu_bwd_acc")
    return dpre, dhp


def _wgrad_into(x, dy, k, dilation, pad_mode, grad, bf16):
    if bf16 and ops.conv_wgrad_bf16_preferred(int(x.shape[1]), int(dy.shape[1]), k, dilation):
        ops.conv_wgrad_bf16(x, dy, k, dilation, pad_mode, out=grad, accumulate=True)
    else:
        ops.conv_wgrad(x, dy, k, dilation, pad_mode, out=grad, accumulate=True)


class _Llg:
    """log_likelihood_gradient and its adjoint for one slice (row-invariant mask): the W = 372 prime-factor kernel when it applies."""

    def __init__(self, blk, y, sense, mask, hybrid):
        self.cfg = (blk.fft_centered, blk.fft_normalization)
        self.sense, self.mask = sense, mask
        if isinstance(hybrid, tuple):
            self.yt, self.op = hybrid
        else:
            self.yt, self.op = hybrid, None
        self.zero_op, self.zero_yt = None, None

    def forward(self, eta, sigma):
        if self.op is not None:
            return ops.llg372(eta, self.op, sigma, self.cfg[1])
        return ops.llg_hinv(eta, self.yt, self.sense, self.mask, sigma, self.cfg[0], self.cfg[1])

    def adjoint(self, dz, sigma):
        """The linear part applied to dz (the map eta -> gradient is affine and self-adjoint: the same kernel with yt = 0)."""
        if self.op is not None:
            if self.zero_op is None:
                self.zero_op = self.op.linear_part()                   # (never reads the data: mrx_llg372 with ytp = NULL)
            return ops.llg372(dz, self.zero_op, sigma, self.cfg[1])
        if self.zero_yt is None:
            self.zero_yt = torch.zeros_like(self.yt)
        return ops.llg_hinv(dz, self.zero_yt, self.sense, self.mask, sigma, self.cfg[0], self.cfg[1])

    def adjoint_parts(self, dz, sigma):
        """The adjoint as (partial planes, their number, 1 / sigma^2) -- W = 372 only (None otherwise): the consumer adds the planes."""
        if self.op is None:
            return None
        if self.zero_op is None:
            self.zero_op = self.op.linear_part()
        parts, n = ops.llg372(dz, self.zero_op, sigma, self.cfg[1], parts=True)
        return parts, n, float(1.0 / (float(sigma) ** 2.0))


def _cascade_forward_backward(blk, eta, llg, tgt, wdev, sigma, bf16):
    """Forward of one RIMBlock cascade (training arithmetic: conv + ReLU and the IndRNN cell as separate launches, their outputs saved),
    its share of the loss, and its backward.  Returns (list of etas, sum of the per-step l1 terms as a device scalar tensor list)."""
    L_ = _lib.lib()
    final = blk.final_layer[0]
    nl = len(blk.layers)
    B, H, W = int(eta.shape[0]), int(eta.shape[1]), int(eta.shape[2])
    plane = H * W
    hx = [None] * nl
    saved, etas, losses = [], [], []
    for _ in range(blk.time_steps):
        g4 = llg.forward(eta, sigma)
        x, acts = g4, []
        for li, st in enumerate(blk.layers):
            c, r = st.convs, st.rnn
            cw, cb = c.conv_layer.weight, c.conv_layer.bias
            use16 = bf16 and ops.conv_bf16_supported(int(cw.shape[1]), int(cw.shape[0]), c.kernel_size, c.dilation)
            a = (ops.conv2d_bf16 if use16 else ops.conv2d)(x, cw, cb, c.dilation, ops.PAD_REPLICATE, ops.ACT_RELU)
            if bf16 and ops.conv_bf16_supported(int(r.ih.weight.shape[1]), int(r.ih.weight.shape[0]), 1, 1):
                h = ops.conv2d_bf16(a, r.ih.weight, r.ih.bias, 1, ops.PAD_ZERO, ops.ACT_RELU, hh=r.hh if hx[li] is not None else None,
                                    h_prev=hx[li])
            else:
                h = ops.indrnn_cell(a, r.ih.weight, r.ih.bias, r.hh, hx[li], 1)
            acts.append((x, a, h, hx[li]))
            hx[li] = h
            x = h
        eta_new = ops.rim_final(x, final.conv_layer.weight, final.conv_layer.bias, final.kernel_size, final.dilation, eta)
        # loss term of this estimate (cirim.py:218-237, l1): mean | target - |eta| / max |eta| |
        m = ops.max_abs(eta_new, complex_modulus=True).reshape(1)
        out2 = torch.empty(2, dtype=torch.float32, device=eta.device)
        work = torch.empty(int(L_.mrx_absl1_work_floats()), dtype=torch.float32, device=eta.device)
        _lib.check(L_.mrx_absl1_loss(_lib.ptr(eta_new), _lib.ptr(tgt), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(work), B * plane, _lib.stream_ptr()),
                   "mrx_absl1_loss")
        saved.append((g4, acts, eta_new, m, out2))
        etas.append(eta_new)
        losses.append(out2)
        eta = eta_new
    # ---- backward, last time-step first -------------------------------------------------------------------------------------------------
    carry, dH = None, [None] * nl
    fw, fb = final.conv_layer.weight, final.conv_layer.bias
    for g4, acts, eta_t, m, out2 in reversed(saved):
        gl = torch.empty_like(eta_t)
        _lib.check(L_.mrx_absl1_loss_bwd(_lib.ptr(eta_t), _lib.ptr(tgt), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(wdev), 1.0, _lib.ptr(gl),
                                         B * plane, _lib.stream_ptr()), "mrx_absl1_loss_bwd")
        tot = torch.empty_like(eta_t)
        d2 = torch.empty(B, 2, H, W, dtype=torch.float32, device=eta_t.device)
        _lib.check(L_.mrx_eta_grad_in(_lib.ptr(carry), _lib.ptr(gl), _lib.ptr(tot), _lib.ptr(d2), B, plane, _lib.stream_ptr()), "mrx_eta_grad_in")
        h_top = acts[-1][2]
        _wgrad_into(h_top, d2, final.kernel_size, final.dilation, ops.PAD_REPLICATE, _grad_of(fw), bf16)
        if fb is not None:
            _grad_of(fb).add_(d2.sum(dim=(0, 2, 3)))                  # two numbers; the model-zoo final conv has no bias
        dh = ag._dgrad(d2, fw, final.dilation, ops.PAD_REPLICATE, bf16 and ops.conv_bf16_supported(2, int(fw.shape[1]), final.kernel_size,
                                                                                                 final.dilation))
        for li in range(nl - 1, -1, -1):
            st = blk.layers[li]
            c, r = st.convs, st.rnn
            x_in, a, h, h_prev = acts[li]
            use16_r = bf16 and ops.conv_bf16_supported(int(r.ih.weight.shape[1]), int(r.ih.weight.shape[0]), 1, 1)
            dpre, dhp = _relu_bwd_acc(dh, dH[li], h, h_prev, r.hh, _grad_of(r.ih.bias) if r.ih.bias is not None else None,
                                      _grad_of(r.hh).reshape(-1) if h_prev is not None else None)
            dH[li] = dhp
            _wgrad_into(a, dpre, 1, 1, ops.PAD_ZERO, _grad_of(r.ih.weight), use16_r)
            da = ag._dgrad(dpre, r.ih.weight, 1, ops.PAD_ZERO, use16_r)
            cw, cb = c.conv_layer.weight, c.conv_layer.bias
            use16_c = bf16 and ops.conv_bf16_supported(int(cw.shape[1]), int(cw.shape[0]), c.kernel_size, c.dilation)
            dpre2, _ = _relu_bwd_acc(da, None, a, None, None, _grad_of(cb) if cb is not None else None, None)
            _wgrad_into(x_in, dpre2, c.kernel_size, c.dilation, ops.PAD_REPLICATE, _grad_of(cw), use16_c)
            dh = ag._dgrad(dpre2, cw, c.dilation, ops.PAD_REPLICATE, use16_c)
        dg4 = dh                                                       # [B,4,H,W]: gradient w.r.t. cat(eta, log-likelihood gradient)
        dz = torch.empty_like(eta_t)
        _lib.check(L_.mrx_g4_to_complex(_lib.ptr(dg4), _lib.ptr(dz), B, plane, _lib.stream_ptr()), "mrx_g4_to_complex")
        t4 = llg.adjoint(dz, sigma)
        carry = torch.empty_like(eta_t)
        _lib.check(L_.mrx_eta_grad_out(_lib.ptr(tot), _lib.ptr(dg4), _lib.ptr(t4), _lib.ptr(carry), B, plane, _lib.stream_ptr()),
                   "mrx_eta_grad_out")
    return etas, losses


def _tl_supported(blk):
    """The bf16-storage tape (csrc/train_bf16.hip) covers the model-zoo RIM: two IndRNN layers of 64 features (5x5 on <= 8 channels, 3x3 dilation 2
    on 64), a final 3x3 convolution into 2 channels without bias."""
    final = blk.final_layer[0]
    if len(blk.layers) != 2 or final is None or final.conv_layer.bias is not None or final.kernel_size != 3 or final.dilation != 1:
        return False
    if tuple(final.conv_layer.weight.shape) != (2, 64, 3, 3):
        return False
    for st in blk.layers:
        c, r = st.convs, st.rnn
        cw = c.conv_layer.weight
        if not ops.tl_layer_supported(int(cw.shape[1]), int(cw.shape[0]), c.kernel_size, c.dilation, int(r.ih.weight.shape[0]), r.kernel_size):
            return False
    return True


def _cascade_forward_backward_tl(blk, eta, llg, tgt, wdev, sigma):
    """One cascade in the reference's mixed-precision arithmetic with bf16 STORAGE (base_cirim_train.yaml:180; what torch.autocast keeps in half
    precision is a pair tensor here): per time-step two fused layer launches + the tap gather forward, and per layer ONE cell-backward launch, one
    data gradient and one weight gradient backward.  Convolution results and the gradients flowing into them are rounded to bf16 exactly where
    autocast rounds them; hidden states, eta, the loss and every parameter-gradient sum are fp32.  The hidden states (and their gradients between
    two cell-backward calls) are channel-blocked [B,8,H,W,8]: five kernels read each of them, all with 16-byte accesses.
    Returns (etas, per-step loss records)."""
    L_ = _lib.lib()
    final = blk.final_layer[0]
    fw = final.conv_layer.weight
    nl = len(blk.layers)
    B, H, W = int(eta.shape[0]), int(eta.shape[1]), int(eta.shape[2])
    plane = H * W
    hx = [None] * nl
    saved, etas, losses = [], [], []
    for _ in range(blk.time_steps):
        g4 = llg.forward(eta, sigma)
        x, acts, taps = g4, [], None
        for li, st in enumerate(blk.layers):
            c, r = st.convs, st.rnn
            a_p, h, taps, hm = ops.tl_layer_fwd(x, c.conv_layer.weight, c.conv_layer.bias, r.ih.weight, r.ih.bias, r.hh, hx[li],
                                                fw if li == nl - 1 else None, want_mask=True)
            acts.append((x, a_p, h, hx[li], hm))
            hx[li] = h
            x = h
        # the estimate of this step, the maximum of its modulus and its l1 term in three launches (gather + per-workgroup maxima, partial sums, final
        # sums): five as separate operators (gather, max partial / final, loss partial / final)
        eta_new = torch.empty_like(eta)
        nmp = int(L_.mrx_tl_final_gather_max_count(B, H, W))
        mp = torch.empty(nmp, dtype=torch.float32, device=eta.device)
        _lib.check(L_.mrx_tl_final_gather_max(_lib.ptr(taps), _lib.ptr(eta), _lib.ptr(eta_new), _lib.ptr(mp), B, H, W, _lib.stream_ptr()),
                   "mrx_tl_final_gather_max")
        m = torch.empty(1, dtype=torch.float32, device=eta.device)
        out2 = torch.empty(2, dtype=torch.float32, device=eta.device)
        work = torch.empty(int(L_.mrx_absl1_work_floats()), dtype=torch.float32, device=eta.device)
        _lib.check(L_.mrx_absl1_loss_mp(_lib.ptr(eta_new), _lib.ptr(tgt), _lib.ptr(mp), nmp, _lib.ptr(m), _lib.ptr(out2), _lib.ptr(work), B * plane,
                                        _lib.stream_ptr()), "mrx_absl1_loss_mp")
        saved.append((acts, eta_new, m, out2))
        etas.append(eta_new)
        losses.append(out2)
        eta = eta_new
    parts = [ops.tl_cell_part(B, H, W, eta.device) for _ in range(nl)]
    main = torch.cuda.current_stream()
    # (inside a hipGraph capture the side stream joins the capture through the events below: the weight gradients become parallel branches of the graph)
    side = _side_stream(eta.device) if TL_SIDE_STREAM and (TL_SIDE_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()) else None

    def on_side(fn, *inputs):
        """fn() on the side stream once everything `main` has queued so far is done; `inputs` were allocated on main and stay alive for it."""
        if side is None:
            return fn()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            fn()
        for t in inputs:
            t.record_stream(side)

    for p in [fw] + [q for st in blk.layers for q in (st.convs.conv_layer.weight,)]:
        _grad_of(p)                                                    # (allocated on the main stream, before any side-stream accumulation)
    carry, dH = None, [None] * nl
    for ti, (acts, eta_t, m, out2) in enumerate(reversed(saved)):
        tot = torch.empty_like(eta_t)
        d2 = torch.empty(B, 2, H, W, dtype=torch.float32, device=eta_t.device)
        _lib.check(L_.mrx_absl1_loss_bwd_eta(_lib.ptr(eta_t), _lib.ptr(tgt), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(wdev), 1.0, _lib.ptr(carry), _lib.ptr(tot),
                                             _lib.ptr(d2), B, plane, _lib.stream_ptr()), "mrx_absl1_loss_bwd_eta")      # (loss backward + mrx_eta_grad_in)
        # final convolution (its result is a bf16 tensor under autocast: both gradient kernels round d2 to bf16 on load)
        h_top = acts[-1][2]
        on_side(lambda h_top=h_top, d2=d2: ops.conv_wgrad_bf16_xcb(h_top, d2, ops.PAD_REPLICATE, out=_grad_of(fw), accumulate=True), h_top, d2)
        dh = ops.tl_dgrad(d2, fw, 1, dx_pairs=True)
        for li in range(nl - 1, -1, -1):
            st = blk.layers[li]
            c, r = st.convs, st.rnn
            x_in, a_p, h, h_prev, hm = acts[li]
            # (the layer's own state enters as its mask bits: 8 bytes per pixel instead of 256)
            dhp, ga = ops.tl_cell_bwd(dh, dH[li], hm, h_prev, a_p, r.ih.weight, fw if li == nl - 1 else None, r.hh, parts[li], ti == 0)
            dH[li] = dhp
            cw = c.conv_layer.weight
            on_side(lambda x_in=x_in, ga=ga, c=c, cw=cw: ops.conv_wgrad_bf16_pairs(x_in, ga, c.kernel_size, c.dilation, ops.PAD_REPLICATE,
                                                                                  out=_grad_of(cw), accumulate=True), x_in, ga)
            dh = ops.tl_dgrad(ga, cw, c.dilation, dx_pairs=li > 0)
        dg4 = dh                                                       # [B,4,H,W] fp32 (bf16 values): gradient w.r.t. cat(eta, log-likelihood gradient)
        dz = torch.empty_like(eta_t)
        _lib.check(L_.mrx_g4_to_complex(_lib.ptr(dg4), _lib.ptr(dz), B, plane, _lib.stream_ptr()), "mrx_g4_to_complex")
        carry = torch.empty_like(eta_t)
        adj = llg.adjoint_parts(dz, sigma)
        if adj is not None:                                            # the adjoint still in its coil-group partial planes: summed by the glue kernel itself
            parts_a, n_a, post = adj
            _lib.check(L_.mrx_eta_grad_out_parts(_lib.ptr(tot), _lib.ptr(dg4), _lib.ptr(parts_a), n_a, post, _lib.ptr(carry), B, plane, _lib.stream_ptr()),
                       "mrx_eta_grad_out_parts")
        else:
            t4 = llg.adjoint(dz, sigma)
            _lib.check(L_.mrx_eta_grad_out(_lib.ptr(tot), _lib.ptr(dg4), _lib.ptr(t4), _lib.ptr(carry), B, plane, _lib.stream_ptr()),
                       "mrx_eta_grad_out")
    if side is not None:
        main.wait_stream(side)                                         # the weight gradients of this cascade are final (its slice may be all-reduced now)
    for li, st in enumerate(blk.layers):                               # the cell kernels' partial sums of the whole cascade -> the gradients
        c, r = st.convs, st.rnn
        ops.tl_cell_reduce(parts[li], B, H, W, _grad_of(r.ih.weight), _grad_of(r.ih.bias) if r.ih.bias is not None else None,
                           _grad_of(r.hh).reshape(-1), _grad_of(c.conv_layer.bias) if c.conv_layer.bias is not None else None)
    return etas, losses


def cirim_forward_backward(model, batch, precision="f32", on_cascade_done=None):
    """Forward, l1 loss (cirim.py:199-247 with accumulate_estimates) and backward of the whole CIRIM on the explicit tape.  Gradients are
    ADDED into `p.grad` of the parameters (zero them first).  `on_cascade_done(i)` is called when cascade i's gradients are final.
    Returns the loss as a 0-dim device tensor."""
    y, S, mask, target = batch["y"], batch["sensitivity_maps"], ops.row_invariant_view(batch["mask"]), batch["target"]
    tgt = target.abs() if target.is_complex() else target
    tgt = (tgt / tgt.abs().max()).abs().float().contiguous()
    T_, nc = model.time_steps, len(model.cirim)
    w = float(torch.logspace(-1, 0, steps=T_).sum()) / T_ / nc          # the reference's weighting quirk (see cirim_l1_loss)
    wdev = torch.full((1,), w, dtype=torch.float32, device=y.device)
    blk0 = model.cirim[0]
    yt = ops.llg_prepare(y, blk0.fft_centered, blk0.fft_normalization, blk0.spatial_dims)
    hybrid = (yt, ops.llg372_prepare(yt, S, mask, blk0.fft_centered, blk0.fft_normalization)) if ops.llg372_supported(yt, mask) else yt
    eta = ops.sens_reduce(y, S, blk0.fft_centered, blk0.fft_normalization, blk0.spatial_dims)     # cascade 0: keep_eta False (rim_block.py:195-211)
    terms = []
    bf16 = precision == "bf16"
    for i, blk in enumerate(model.cirim):
        llg = _Llg(blk, y, S, mask, hybrid)
        if bf16 and BF16_STORAGE and _tl_supported(blk):
            etas, losses = _cascade_forward_backward_tl(blk, eta, llg, tgt, wdev, 1.0)
        else:
            etas, losses = _cascade_forward_backward(blk, eta, llg, tgt, wdev, 1.0, bf16)
        terms += losses
        eta = etas[-1]                                                    # keep_eta: the next cascade starts from pred[-1].detach()
        if on_cascade_done is not None:
            on_cascade_done(i)
    total = torch.stack(terms)[:, 0].sum() * w
    return total


def training_step(model, flat, optimizer, batch, time_steps=None, schedule=None, use_tape=None):
    """One data-parallel step: forward (recorded), l1 loss, backward through the HIP kernels, ONE all-reduce of the flat gradient,
    Adam.  `batch`: dict with y, sensitivity_maps, mask, target.  `schedule`: optional dict(max_steps, base_lr, warmup_ratio | warmup_steps,
    min_lr) -- the learning rate of this step is then the reference's InverseSquareRootAnnealing value (inverse_sqrt_lr at the number of
    optimizer steps taken so far).  Returns the loss (0-dim device tensor)."""
    model.train()
    if schedule is not None:
        optimizer.lr = inverse_sqrt_lr(optimizer.steps, schedule["max_steps"], schedule["base_lr"], schedule.get("warmup_ratio", 0.1),
                                       schedule.get("min_lr", 0.0), schedule.get("warmup_steps"))
    flat.zero_grad()
    if use_tape is None:
        use_tape = _tape_supported(model, batch)
    if use_tape:
        # explicit tape: each cascade's slice of the flat gradient is final when its backward ends -- its all-reduce is issued right
        # there (async) and overlaps the next cascade; everything is waited for before the optimizer
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        handles = []

        def reduce_cascade(i):
            if multi:
                o0, o1 = flat.cascade_slices[i]
                handles.append(dist.all_reduce(flat.grad[o0:o1], op=dist.ReduceOp.SUM, async_op=True))

        loss = cirim_forward_backward(model, batch, ag.PRECISION, reduce_cascade if flat.cascade_slices else None)
        if multi:
            if not flat.cascade_slices:
                handles.append(dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, async_op=True))
            else:
                for o0, o1 in flat.rest_slices:
                    handles.append(dist.all_reduce(flat.grad[o0:o1], op=dist.ReduceOp.SUM, async_op=True))
            for h in handles:
                h.wait()
        world = dist.get_world_size() if multi else 1
        optimizer.step(grad_scale=1.0 / world)
        return loss.detach()
    etas = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
    loss = cirim_l1_loss(etas, batch["target"], model.time_steps, len(model.cirim))
    loss.backward()
    world = allreduce_gradients(flat.grad)
    optimizer.step(grad_scale=1.0 / world)
    return loss.detach()


def magnitude_l1_loss(pred, target):
    """mean | |pred| / max(target) - target / max(target) | on a complex prediction [B,h,w] (the l1 branch of the reference's
    process_loss for a single estimate, vn.py training_step -> base.py:188-254)."""
    t = torch.abs(target / torch.max(torch.abs(target)))
    p = torch.abs(pred) / torch.max(torch.abs(target))
    return (p - t).abs().mean()


def model_training_step(model, flat, optimizer, batch, loss_fn=magnitude_l1_loss, schedule=None):
    """One data-parallel step of a model whose forward is recorded through `mridc_amd.diff` (E2EVN / VarNet, UNet, RIMs with gated
    cells): forward, loss, backward on the HIP kernels, one all-reduce of the flat gradient, Adam.  Returns the loss."""
    model.train()
    if schedule is not None:
        optimizer.lr = inverse_sqrt_lr(optimizer.steps, schedule["max_steps"], schedule["base_lr"], schedule.get("warmup_ratio", 0.1),
                                       schedule.get("min_lr", 0.0), schedule.get("warmup_steps"))
    flat.zero_grad()
    pred = model(batch["y"], batch["sensitivity_maps"], batch["mask"], batch.get("init_pred"), batch["target"])
    loss = loss_fn(pred, batch["target"])
    loss.backward()
    world = allreduce_gradients(flat.grad)
    optimizer.step(grad_scale=1.0 / world)
    return loss.detach()


class GraphedModelStep:
    """`model_training_step` with forward + loss + backward captured into ONE hipGraph (the step of an E2EVN-sized model is a few thousand launches of
    10-50 us kernels: issued from Python it is host-bound and jitters with the host; replayed it is not).  The batch lives in static buffers that every
    call refills; the all-reduce of the flat gradient and the Adam launch stay outside the graph (the step count and the learning rate are host values).
    Weight packs are made INSIDE the graph (ops.WEIGHTS_DYNAMIC): every replay packs the weights the optimizer has just updated.

    The forward must be free of host synchronisation and of data-dependent Python branches (true of the modules recorded through mridc_amd.diff);
    shapes and the mask layout are fixed at construction."""

    def __init__(self, model, flat, optimizer, batch, loss_fn=magnitude_l1_loss, warmup=2):
        self.model, self.flat, self.optimizer, self.loss_fn = model, flat, optimizer, loss_fn
        self.keys = [k for k in ("y", "sensitivity_maps", "mask", "init_pred", "target") if batch.get(k) is not None]
        self.static = {k: batch[k].clone() for k in self.keys}
        model.train()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(max(int(warmup), 1)):       # eager warm-up on the capture stream (allocator, autograd buffers, lazily built caches)
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(st)
        self.graph = torch.cuda.CUDAGraph()
        keep = ops.WEIGHTS_DYNAMIC
        ops.WEIGHTS_DYNAMIC = True
        try:
            with torch.cuda.graph(self.graph, stream=st, capture_error_mode="thread_local"):
                self.loss = self._fwd_bwd()
        finally:
            ops.WEIGHTS_DYNAMIC = keep

    def _fwd_bwd(self):
        b = self.static
        self.flat.zero_grad()
        pred = self.model(b["y"], b["sensitivity_maps"], b["mask"], b.get("init_pred"), b["target"])
        loss = self.loss_fn(pred, b["target"])
        loss.backward()
        return loss.detach()

    def __call__(self, batch, schedule=None):
        if schedule is not None:
            self.optimizer.lr = inverse_sqrt_lr(self.optimizer.steps, schedule["max_steps"], schedule["base_lr"], schedule.get("warmup_ratio", 0.1),
                                                schedule.get("min_lr", 0.0), schedule.get("warmup_steps"))
        missing = [k for k in self.keys if k not in batch]
        if missing:
            raise KeyError(f"GraphedModelStep: the captured step reads {sorted(self.keys)}; this batch lacks {missing} (capture a new step for another batch layout)")
        for k in self.keys:
            if batch[k] is not self.static[k]:
                self.static[k].copy_(batch[k])
        self.graph.replay()
        world = allreduce_gradients(self.flat.grad)
        self.optimizer.step(grad_scale=1.0 / world)
        return self.loss.clone()           # (self.loss lives in the graph's pool: the next replay overwrites it)


class GraphedCirimStep(GraphedModelStep):
    """`training_step` on the explicit tape (cirim_forward_backward) as one hipGraph replay; the weight gradients of the bf16-storage tape are parallel
    branches of the graph (the side stream joins the capture).  One all-reduce of the whole flat gradient after the replay (the eager step overlaps
    per-cascade all-reduces with the following cascades instead), then Adam."""

    def __init__(self, model, flat, optimizer, batch, precision=None, warmup=2):
        self.precision = ag.PRECISION if precision is None else precision
        if not _tape_supported(model, batch):
            raise NotImplementedError("GraphedCirimStep: the explicit tape does not cover this model / mask")
        super().__init__(model, flat, optimizer, batch, None, warmup)

    def _fwd_bwd(self):
        self.flat.zero_grad()
        return cirim_forward_backward(self.model, self.static, self.precision, None).detach()
