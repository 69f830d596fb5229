"""Oracle: the reference's mixed-precision training arithmetic (BASELINE config 4).  Test infrastructure.

The reference trains under pytorch-lightning's native AMP (`precision: 16`, projects/reconstruction/model_zoo/conf/base_cirim_train.yaml:180;
mridc/core/... hands the trainer block to PTL 1.7.7, whose NativeMixedPrecisionPlugin wraps forward + loss in `torch.autocast`): convolutions
run on half-precision operands and RETURN half-precision tensors, everything autocast does not list (FFT, the complex products of
log_likelihood_gradient, `hh * hx`, the loss) stays fp32.  Two restatements of that arithmetic over the fp32 oracle, used as checkers of the
HIP bf16 tape (`mridc_amd.training.cirim_forward_backward(..., "bf16")`):

* `autocast_bf16()`  -- the oracle under `torch.autocast("cpu", dtype=torch.bfloat16)`: torch's own autocast rules (what `precision: bf16` gives the
  reference; fp16 + loss scaling differs from it only by the width of the significand).  Conv outputs and the gradients flowing into them are
  rounded to bf16, and the gradient of a weight that is used at several time-steps is accumulated in bf16 (autocast caches the cast weight).
* `bf16_operand_convs()` -- the arithmetic of the HIP kernels themselves: every convolution (forward, data gradient, weight gradient) rounds its
  two OPERANDS to bf16 (round to nearest even), multiplies exactly and accumulates wide (float64 here, fp32 in the MFMA), results and stored
  tensors stay fp32.  This is the tight checker: it differs from the kernels only by the order of the fp32 accumulation.

`autocast` vs the fp32 oracle differ by 1e-2 .. 4e-2 in the whole gradient vector (tests/test_oracle_amp.py pins that on the CPU): a bf16
implementation cannot be checked against the fp32 oracle more tightly than that, which is why these two exist."""
import contextlib

import torch
import torch.nn.functional as F

from . import rim as orim


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding, dilation, round_forward, round_results):
        ctx.save_for_backward(x, w)
        ctx.cfg = (padding, dilation, b is not None, round_results)
        r = bf16_round if round_forward else (lambda t: t)
        y = F.conv2d(r(x).double(), r(w).double(), None, padding=padding, dilation=dilation).float()
        if round_results:                                                  # the convolution RETURNS bf16 (autocast): bias joins before the rounding
            return bf16_round(y + bf16_round(b).view(1, -1, 1, 1) if b is not None else y)
        return y + b.view(1, -1, 1, 1) if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        padding, dilation, has_bias, round_results = ctx.cfg
        dyb = bf16_round(dy).double()
        dx = torch.nn.grad.conv2d_input(x.shape, bf16_round(w).double(), dyb, padding=padding, dilation=dilation).float() if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(bf16_round(x).double(), w.shape, dyb, padding=padding, dilation=dilation).float() if ctx.needs_input_grad[1] else None
        if round_results:                                                  # gradient w.r.t. a bf16 input is bf16; weight / bias gradients accumulate in fp32
            dx = bf16_round(dx) if dx is not None else None
            db = dyb.sum((0, 2, 3)).float() if has_bias else None
        else:
            db = dy.sum((0, 2, 3)) if has_bias else None                   # the fp32-storage kernels sum the fp32 gradient (mrx_relu_bwd_acc)
        return dx, dw, db, None, None, None, None


def _conv2d_bf16_operands(x, weight, bias=None, padding=0, dilation=1, round_forward=True, round_results=False):
    return _ConvBf16Operands.apply(x, weight, bias, padding, dilation, round_forward, round_results)


@contextlib.contextmanager
def bf16_operand_convs(skip=(), fp32_forward=(), round_results=False):
    """Inside: the RIM's 2-D convolutions (oracle.rim) multiply bf16-rounded operands.  `skip`: (Cin, Cout) pairs that stay fp32 (layers the
    product keeps on its fp32 kernels); `fp32_forward`: pairs whose FORWARD stays fp32 while both gradients round their operands;
    `round_results`: the convolutions also RETURN bf16-representable values (forward result with its bias, data gradient) as autocast's do,
    while weight / bias gradients are summed in fp32 -- the arithmetic of the HIP tape with bf16 storage."""
    keep = orim._CONV2D[0]

    def conv(x, weight, bias=None, padding=0, dilation=1):
        io = (int(weight.shape[1]), int(weight.shape[0]))
        if io in skip:
            return keep(x, weight, bias, padding=padding, dilation=dilation)
        return _conv2d_bf16_operands(x, weight, bias, padding, dilation, io not in fp32_forward, round_results)
    orim._CONV2D[0] = conv
    try:
        yield
    finally:
        orim._CONV2D[0] = keep


def autocast_bf16():
    """The reference's AMP context on the CPU (bf16: the half type torch's CPU autocast implements)."""
    return torch.autocast("cpu", dtype=torch.bfloat16)


# ---- precision-16 INFERENCE (base_cirim_run.yaml:132 `precision: 16`; the checkers of csrc/rim_amp16.hip) ---------------------------------------------
def autocast_fp16():
    """The reference's inference arithmetic: pytorch-lightning's native AMP for `precision: 16` is `torch.autocast(dtype=torch.float16)` around the
    forward pass.  torch's CPU autocast implements float16 with the same op lists (convolutions in half precision RETURNING half precision; FFT, complex
    products, `hh * hx`, eta stay fp32)."""
    return torch.autocast("cpu", dtype=torch.float16)


def fp16_round(x):
    return x.to(torch.float16).to(torch.float32)


@contextlib.contextmanager
def fp16_kernel_arithmetic():
    """The arithmetic of the precision-16 KERNELS (mrx_amp16_layer1 / _layer2) restated on the CPU: every 2-D convolution of the regulariser multiplies
    fp16-rounded operands exactly and accumulates wide (float64 here, fp32 in the MFMA) with an fp32 result -- the rounding of ReLU(conv + b) to fp16 before
    the 1x1 cell IS that cell's operand rounding -- and a recurrent cell's new state is rounded to fp16 once (the kernels keep their states in fp16;
    autocast keeps them fp32 but rounds every convolution's OUTPUT, which the kernels do not).  Differs from the kernels only by the order of the fp32 sums
    and by the first layer's exact power-of-two input scale (no fp16 underflow of small inputs there)."""
    keep_conv, keep_state = orim._CONV2D[0], orim._STATE[0]

    def conv(x, weight, bias=None, padding=0, dilation=1):
        y = F.conv2d(fp16_round(x).double(), fp16_round(weight).double(), None, padding=padding, dilation=dilation)
        return (y + bias.double().view(1, -1, 1, 1) if bias is not None else y).float()
    from . import unet as ounet           # ... and of mrx_unet_conv3x3_p16: the 3x3 convolutions of the U-Net blocks on fp16-rounded operands, fp32 results
    keep_unet = ounet._CONV3X3[0]
    orim._CONV2D[0], orim._STATE[0], ounet._CONV3X3[0] = conv, fp16_round, conv
    try:
        yield
    finally:
        orim._CONV2D[0], orim._STATE[0], ounet._CONV3X3[0] = keep_conv, keep_state, keep_unet


def cirim_loss_and_gradients(state, cfg, sample, mode="fp32", skip=(), fp32_forward=(), round_results=False):
    """Forward + l1 loss (cirim.py:199-247) + torch autograd of the CIRIM oracle in one of the three arithmetics (`fp32`, `autocast_bf16`,
    `bf16_operands`).  `state`: reference state_dict (fp32 tensors); returns (loss, {name: gradient})."""
    from . import models as omodels
    p = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()}
    ctx = {"fp32": contextlib.nullcontext, "autocast_bf16": autocast_bf16, "bf16_operands": lambda: bf16_operand_convs(skip, fp32_forward, round_results)}[mode]()
    with ctx:
        pred = omodels.cirim_forward(p, cfg, sample["y"], sample["sensitivity_maps"], sample["mask"], None, sample["target"])
        loss = omodels.cirim_process_loss(sample["target"], pred, torch.nn.L1Loss(), omodels.cirim_time_steps(cfg["time_steps"]), cfg["num_cascades"])
    loss.backward()
    return loss.detach().float(), {k: v.grad.float() for k, v in p.items() if v.grad is not None}


# ---- tie-free targets for gradient comparisons ------------------------------------------------------------------------------------------------------
def detie_terms(t, vals, margin, lo=None, hi=None):
    """Copy of the real tensor `t` in which no element is within `margin` of the same element of any tensor stacked in `vals` [E, *t.shape]
    (only the tied elements move, in steps of `margin`, staying inside (lo, hi))."""
    t = t.detach().clone()
    bad = ((vals - t.unsqueeze(0)).abs() < margin).any(0)
    for idx in bad.nonzero().tolist():
        idx = tuple(idx)
        col, cur = vals[(slice(None),) + idx], float(t[idx])
        for k in range(1, 400):
            cands = [c for c in (cur - k * margin, cur + k * margin) if (lo is None or c > lo) and (hi is None or c < hi)
                     and float((col - c).abs().min()) >= margin]
            if cands:
                t[idx] = cands[0]
                break
        else:
            raise AssertionError("no tie-free value for an element of the target")
    assert float((vals - t.unsqueeze(0)).abs().min()) >= 0.5 * margin
    return t


def detie_l1_target(target, preds, margin=1e-3):
    """Gradient tests of the reference's l1 loss (cirim.py:218-237: mean | t / max t - |p| / max |p| |) compare two fp32 implementations whose
    forward results differ by ~1e-7; the derivative of a term is its SIGN, so a term within round-off of zero -- or two pixels tying for max |p| --
    makes the comparison a coin toss that says nothing about the kernels.  Returns a copy of `target` (real, >= 0, [B,H,W]) in
    which every term of every estimate in `preds` (list of lists of complex [B,H,W], the oracle's forward) is at least `margin` away from zero,
    moving only the tied pixels and never the maximum; asserts that every estimate's largest modulus leads the runner-up by 1e-4 (relative)."""
    t = target.detach().clone().float()
    tmax = float(t.abs().max())
    pn = []
    for cascade in preds:
        for p in cascade:
            a = p.detach().abs().float()
            top = torch.topk(a.reshape(-1), 2).values
            assert float(top[0] - top[1]) > 1e-4 * float(top[0]), "two pixels tie for max |p|: pick another slice for this test"
            pn.append(a / top[0])
    out = detie_terms((t / tmax).abs(), torch.stack(pn), margin, lo=0.0, hi=1.0 - margin) * tmax
    out = torch.where(out == out, out, t)
    keep = (t / tmax).abs() >= 1.0 - margin                 # the maximum (and anything that close to it) stays exactly what it was
    out = torch.where(keep, t, out)
    assert float(out.abs().max()) == tmax
    return out
