"""Training path (BASELINE config C4): torch.autograd.Function wrappers whose forward AND backward are the HIP kernels
of this package.  The reference trains the CIRIM through torch autograd (cuDNN / MIOpen backward of every op of
rim_block.py:217-249); here autograd only keeps the tape -- the arithmetic of every backward step is a launch of
libmridc_amd.so (mrx_conv_wgrad, the forward conv kernels as data gradient + mrx_reppad_fold, mrx_relu_bwd, the
log-likelihood-gradient kernel as its own adjoint, mrx_absl1_loss_bwd).

`RIMBlock.forward` takes this path in train() mode when gradients are enabled and a parameter requires one; the inference path (fused layer
kernels, no saved activations) is untouched."""
import torch

from mridc_amd import _lib, ops

# Arithmetic of the convolutional regulariser on the training path: "f32" (fp32 matrix cores, the inference kernels' arithmetic) or
# "bf16" -- BASELINE config 4: bf16 operands with fp32 accumulation for every convolution / IndRNN GEMM, forward and backward (the
# reference trains under AMP, base_cirim_train.yaml:180), while the FFTs, the data consistency, the eta accumulation, the loss and the
# optimizer stay fp32.  Set through `set_precision`; each Function records the mode it ran its forward in.
PRECISION = "f32"


def set_precision(mode):
    global PRECISION
    if mode not in ("f32", "bf16"):
        raise ValueError(f"precision must be 'f32' or 'bf16', got {mode!r}")
    PRECISION = mode


def _bf16(cin, cout, k, dilation):
    return PRECISION == "bf16" and ops.conv_bf16_supported(cin, cout, k, dilation)


def _dgrad(dy, weight, dilation, pad_mode, bf16):
    """Data gradient of y = conv(pad(x), weight).  bf16: the zero-padded convolution with the flipped, transposed weights on bf16 operands
    (mrx_conv2d_bf16 with a transposed pack), folded back by mrx_reppad_fold for replicate padding."""
    if not bf16:
        return ops.conv_dgrad(dy, weight, dilation, pad_mode)
    k = int(weight.shape[-1])
    pad = int(dilation) * (k - 1) // 2
    if pad_mode == ops.PAD_ZERO or pad == 0:
        return ops.conv2d_bf16(dy, weight, None, dilation, ops.PAD_ZERO, transposed=True)
    # replicate padding: interior of the gradient straight into dx, its frame into a side buffer, then the edge pixels (2 launches; the
    # frame buffer's interior is never written or read)
    dy = _lib.f32c(dy)
    B, Cout_w, H, W = int(dy.shape[0]), int(weight.shape[0]), int(dy.shape[2]), int(dy.shape[3])
    Cin = int(weight.shape[1])
    packed = ops._conv_bf16_pack(weight, True)
    out = torch.empty(B, Cin, H, W, dtype=torch.float32, device=dy.device)
    frame = torch.empty(B, Cin, H + 2 * pad, W + 2 * pad, dtype=torch.float32, device=dy.device)
    L = _lib.lib()
    _lib.check(L.mrx_conv2d_bf16_dgrad_rep(_lib.ptr(dy), _lib.ptr(packed), _lib.ptr(out), _lib.ptr(frame), B, Cout_w, Cin, H, W, k, int(dilation),
                                           _lib.stream_ptr()), "mrx_conv2d_bf16_dgrad_rep")
    _lib.check(L.mrx_reppad_fold_edges(_lib.ptr(frame), _lib.ptr(out), B * Cin, H, W, pad, _lib.stream_ptr()), "mrx_reppad_fold_edges")
    return out


def _wgrad(x, dy, k, dilation, pad_mode, bf16):
    if bf16 and ops.conv_wgrad_bf16_preferred(int(x.shape[1]), int(dy.shape[1]), k, dilation):
        return ops.conv_wgrad_bf16(x, dy, k, dilation, pad_mode)
    return ops.conv_wgrad(x, dy, k, dilation, pad_mode)


_ZEROS = {}


def _zeros_like(t):
    """A shared all-zero buffer of t's shape (the `y = 0` operand of the adjoint gradient; never written)."""
    key = (tuple(t.shape), str(t.device))
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) >= 8:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros_like(t)
    return z


class ConvReLU(torch.autograd.Function):
    """g = ReLU(conv_reppad(x; w, b))  (conv_layers.py:121-123)."""

    @staticmethod
    def forward(ctx, x, w, b, dilation):
        ctx.bf16 = _bf16(int(w.shape[1]), int(w.shape[0]), int(w.shape[-1]), int(dilation))
        if ctx.bf16:
            g = ops.conv2d_bf16(x, w, b, dilation, ops.PAD_REPLICATE, ops.ACT_RELU)
        else:
            g = ops.conv2d(x, w, b, dilation, ops.PAD_REPLICATE, ops.ACT_RELU)
        ctx.save_for_backward(x, w, g)
        ctx.dilation, ctx.has_bias = int(dilation), b is not None
        return g

    @staticmethod
    def backward(ctx, dg):
        x, w, g = ctx.saved_tensors
        dpre, _, sums = ops.relu_bwd(dg, g)
        dw = _wgrad(x, dpre, int(w.shape[-1]), ctx.dilation, ops.PAD_REPLICATE, ctx.bf16)
        dx = _dgrad(dpre, w, ctx.dilation, ops.PAD_REPLICATE, ctx.bf16) if ctx.needs_input_grad[0] else None
        return dx, dw, (sums[:, 0].contiguous() if ctx.has_bias else None), None


class IndRNN1x1(torch.autograd.Function):
    """h = ReLU(W_ih g + b_ih + hh * h_prev) with a 1x1 `ih` (rnn_cells.py:384-391)."""

    @staticmethod
    def forward(ctx, g, w_ih, b_ih, hh, h_prev):
        ctx.bf16 = _bf16(int(w_ih.shape[1]), int(w_ih.shape[0]), 1, 1)
        if ctx.bf16:
            h = ops.conv2d_bf16(g, w_ih, b_ih, 1, ops.PAD_ZERO, ops.ACT_RELU, hh=hh if h_prev is not None else None, h_prev=h_prev)
        else:
            h = ops.indrnn_cell(g, w_ih, b_ih, hh, h_prev, 1)
        ctx.save_for_backward(g, w_ih, hh, h_prev if h_prev is not None else torch.empty(0, device=g.device), h)
        ctx.has_bias, ctx.has_prev = b_ih is not None, h_prev is not None
        return h

    @staticmethod
    def backward(ctx, dh):
        g, w_ih, hh, h_prev, h = ctx.saved_tensors
        hp = h_prev if ctx.has_prev else None
        dpre, dhp, sums = ops.relu_bwd(dh, h, hp, hh if ctx.has_prev else None)
        dw = _wgrad(g, dpre, 1, 1, ops.PAD_ZERO, ctx.bf16)
        dg = _dgrad(dpre, w_ih, 1, ops.PAD_ZERO, ctx.bf16)
        dhh = sums[:, 1].reshape(hh.shape).contiguous() if ctx.has_prev else torch.zeros_like(hh)
        return dg, dw, (sums[:, 0].contiguous() if ctx.has_bias else None), dhh, dhp


class RimFinal(torch.autograd.Function):
    """eta_new = eta + permute(conv_reppad(h; w, b), (0, 2, 3, 1))  (rim_block.py:239-248)."""

    @staticmethod
    def forward(ctx, h, w, b, dilation, eta):
        # the 64 -> 2 convolution and the eta accumulation stay fp32 in both modes (a VALU / HBM-bound layer: 0.55 of 25 GFLOP); its
        # data gradient (2 -> 64 channels) runs on the bf16 matrix cores in bf16 mode
        ctx.bf16 = _bf16(2, int(w.shape[1]), int(w.shape[-1]), int(dilation))
        out = ops.rim_final(h, w, b, int(w.shape[-1]), dilation, eta)
        ctx.save_for_backward(h, w)
        ctx.dilation, ctx.has_bias = int(dilation), b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        h, w = ctx.saved_tensors
        d2 = dout.permute(0, 3, 1, 2).contiguous()                      # [B,2,H,W]
        dw = _wgrad(h, d2, int(w.shape[-1]), ctx.dilation, ops.PAD_REPLICATE, ctx.bf16)
        dh = _dgrad(d2, w, ctx.dilation, ops.PAD_REPLICATE, ctx.bf16)
        db = d2.sum(dim=(0, 2, 3)) if ctx.has_bias else None            # 2 numbers (the model-zoo final conv has no bias)
        return dh, dw, db, None, dout


class LogLikelihoodGradient(torch.autograd.Function):
    """cat(eta, grad) with grad = sum_c conj(S) F^-1 (mask (F(S eta) - y)) / sigma^2  (rim_utils.py:11-67).  The map eta -> grad is
    affine with a self-adjoint linear part (F^H = N F^-1 cancels the normalisation), so the backward pass is the forward kernel
    applied to the incoming gradient with y = 0."""

    @staticmethod
    def forward(ctx, eta, y_or_yt, sens, mask, sigma, centered, normalization, hinv):
        if hinv:
            out = ops.llg_hinv(eta, y_or_yt, sens, mask, sigma, centered, normalization)
        else:
            out = ops.llg(eta, y_or_yt, sens, mask, sigma, centered, normalization)
        ctx.save_for_backward(sens, mask)
        ctx.cfg = (float(sigma), bool(centered), normalization, bool(hinv))
        return out

    @staticmethod
    def backward(ctx, dout):
        sens, mask = ctx.saved_tensors
        sigma, centered, normalization, hinv = ctx.cfg
        dz = dout[:, 2:4].permute(0, 2, 3, 1).contiguous()              # gradient w.r.t. the `grad` channels, as a complex image
        zero = _zeros_like(sens)
        if hinv:
            t = ops.llg_hinv(dz, zero, sens, mask, sigma, centered, normalization)
        else:
            t = ops.llg(dz, zero, sens, mask, sigma, centered, normalization)
        one = torch.ones(1, dtype=torch.float32, device=dout.device)
        deta = ops.lincomb(dout[:, 0:2].contiguous(), t[:, 2:4].contiguous(), one, 2)   # d/d eta of cat(eta, .) + adjoint part
        return deta.permute(0, 2, 3, 1).contiguous(), None, None, None, None, None, None, None


class AbsL1Loss(torch.autograd.Function):
    """mean |target - |p| / max|p||  for one complex prediction p [..., 2] (cirim.py:218-237 with l1)."""

    @staticmethod
    def forward(ctx, p, target):
        p, target = _lib.f32c(p), _lib.f32c(target)
        n = p.numel() // 2
        if target.numel() != n:
            raise ValueError(f"AbsL1Loss: prediction {tuple(p.shape)} vs target {tuple(target.shape)}")
        L = _lib.lib()
        m = ops.max_abs(p, complex_modulus=True).reshape(1)
        out2 = torch.empty(2, dtype=torch.float32, device=p.device)
        work = torch.empty(int(L.mrx_absl1_work_floats()), dtype=torch.float32, device=p.device)
        _lib.check(L.mrx_absl1_loss(_lib.ptr(p), _lib.ptr(target), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(work), n, _lib.stream_ptr()),
                   "mrx_absl1_loss")
        ctx.save_for_backward(p, target, m, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, gout):
        p, target, m, out2 = ctx.saved_tensors
        dp = torch.empty_like(p)
        g = _lib.f32c(gout.reshape(1))                                   # upstream gradient stays on the device (no host read)
        _lib.check(_lib.lib().mrx_absl1_loss_bwd(_lib.ptr(p), _lib.ptr(target), _lib.ptr(m), _lib.ptr(out2), _lib.ptr(g), 1.0, _lib.ptr(dp),
                                                 p.numel() // 2, _lib.stream_ptr()), "mrx_absl1_loss_bwd")
        return dp, None
