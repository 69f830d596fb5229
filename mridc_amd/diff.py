"""Differentiable forms of the operators the NormUnet / VarNetBlock / gated-cell forward passes call (training under the
reference's trainer covers E2EVN, models/vn.py:94-142, and RIMs with GRU / MGU cells, rim_block.py:217-249, beside CIRIM).

Same names and arguments as `mridc_amd.ops`; a module picks this namespace instead of `ops` when gradients are being recorded
(`diff.active(...)`).  Everything heavy stays on the HIP kernels in both directions: convolutions (forward `mrx_conv2d` / Winograd, data
gradient `mrx_conv2d` on flipped weights with the replicate-padding fold, weight gradient `mrx_conv_wgrad` -- the generic matrix-core
kernel for arbitrary channel counts), the transposed 2x2 convolution (its gradients are 1x1 convolutions of the pixel-unshuffled output
gradient), FFTs (the adjoint of a transform is the opposite transform times N^(+-1) for the unnormalised conventions), instance norm, group norm with
its statistics' gradients and the un-normalisation (`mrx_group_norm_bwd`), zero padding / cropping (`mrx_pad2d` both ways), pooling, activations'
derivatives, the GRU / MGU gate math (`mrx_gru_gates_bwd` / `mrx_mgu_gates_bwd`).  What torch still records: channel concatenation (its backward is two
views), the one-pixel reflect padding of odd sizes in the U-Net's up path, and the pointwise complex arithmetic of the coil operators' glue."""
import torch
import torch.nn.functional as F

import mridc_amd.collections.common.parts.fft as fft
from mridc_amd import ops

PAD_ZERO, PAD_REPLICATE = ops.PAD_ZERO, ops.PAD_REPLICATE
ACT_NONE, ACT_RELU, ACT_LEAKY = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY


def active(*tensors, training=True):
    """True when torch is recording gradients and one of the tensors takes part.  Modules pass `training=self.training`: an eval-mode module
    called without torch.no_grad() on inputs that need no gradient takes the fused inference kernels instead of silently recording a (several
    times slower) tape -- the same rule RIMBlock.forward applies."""
    if not torch.is_grad_enabled():
        return False
    req = [t for t in tensors if torch.is_tensor(t) and t.requires_grad]
    if not req:
        return False
    # eval(): parameters alone do not switch the tape on (inference without torch.no_grad()), but an INPUT that requires grad does -- a frozen
    # eval-mode sub-network inside a trained pipeline, test-time adaptation, saliency: eval() does not disable autograd, and the fused kernels'
    # outputs carry no grad_fn
    return bool(training) or any(not isinstance(t, torch.nn.Parameter) for t in req)


def _act_grad(dy, y, act, slope):
    return ops.act_bwd(dy, y, act, slope)                  # dy * act'(y): mrx_act_bwd


class _Conv2d(torch.autograd.Function):
    """act(conv(pad(x), w) + b), 'same' size."""

    @staticmethod
    def forward(ctx, x, w, b, dilation, pad_mode, act, slope):
        y = ops.conv2d(x, w, b, dilation, pad_mode, act, slope)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (int(dilation), int(pad_mode), int(act), float(slope), b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dilation, pad_mode, act, slope, has_b = ctx.cfg
        g = _act_grad(dy.contiguous(), y, act, slope)
        k = int(w.shape[-1])
        dx = ops.conv_dgrad(g, w, dilation, pad_mode) if ctx.needs_input_grad[0] else None
        dw = ops.conv_wgrad(x, g, k, dilation, pad_mode) if ctx.needs_input_grad[1] else None
        db = g.sum((0, 2, 3)) if (has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


def conv2d(x, weight, bias, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, out=None):
    return _Conv2d.apply(x, weight, bias, dilation, pad_mode, act, slope)


class _InstanceNormAct(torch.autograd.Function):
    """act((x - mean) / sqrt(var + eps)) per (b, c) plane, biased variance (InstanceNorm2d without affine)."""

    @staticmethod
    def forward(ctx, x, eps, act, slope):
        if act not in (ACT_NONE, ACT_LEAKY):
            raise NotImplementedError("instance norm + ReLU backward (the normalised value is not recoverable from the output)")
        out, work = ops.instance_norm_act(x, eps, act, slope, inplace=False, return_work=True)
        ctx.save_for_backward(out, work)                   # work: the forward's per-plane partial sums (rstd comes from them in the backward kernel)
        ctx.cfg = (float(eps), int(act), float(slope))
        return out

    @staticmethod
    def backward(ctx, dy):
        out, work = ctx.saved_tensors
        eps, act, slope = ctx.cfg
        return ops.instance_norm_act_bwd(dy, out, work, eps, act, slope), None, None, None      # mrx_inorm_act_bwd: two passes, no torch arithmetic


def instance_norm_act(x, eps=1e-5, act=ACT_LEAKY, slope=0.2, inplace=True):
    return _InstanceNormAct.apply(x, eps, act, slope)


def conv_instance_norm_act(x, weight, eps=1e-5, act=ACT_LEAKY, slope=0.2, pad_mode=PAD_ZERO):
    return instance_norm_act(conv2d(x, weight, None, 1, pad_mode), eps, act, slope)


class _ConvTranspose2x2(torch.autograd.Function):
    """ConvTranspose2d(kernel 2, stride 2, no bias): out[b,co,2h+i,2w+j] = sum_ci x[b,ci,h,w] w[ci,co,i,j]."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.conv_transpose2x2(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        cin, cout = int(w.shape[0]), int(w.shape[1])
        dyu = ops.pixel_unshuffle2(dy)                                   # [B, cout * 4, H, W], channel (co, i, j): mrx_pixel_unshuffle2
        dx = ops.conv2d(dyu, w.detach().reshape(cin, cout * 4, 1, 1), None) if ctx.needs_input_grad[0] else None
        dw = ops.conv_wgrad(dyu, x, 1, 1, PAD_ZERO).reshape(cin, cout, 2, 2) if ctx.needs_input_grad[1] else None
        return dx, dw


def conv_transpose2x2(x, weight):
    return _ConvTranspose2x2.apply(x, weight)


class _AvgPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (int(x.shape[-2]), int(x.shape[-1]))
        return ops.avg_pool2x2(x)

    @staticmethod
    def backward(ctx, dy):
        H, W = ctx.hw
        return ops.avg_pool2x2_bwd(dy, H, W)                             # mrx_avgpool2x2_bwd (an odd last row / column is not pooled: 0)


def avg_pool2x2(x):
    return _AvgPool2x2.apply(x)


class _ZeroPad2d(torch.autograd.Function):
    """Zero padding / cropping (negative pads) of the last two dims: mrx_pad2d in both directions -- the gradient of a pad is the opposite crop and vice versa."""

    @staticmethod
    def forward(ctx, x, top, bottom, left, right):
        ctx.pads = (int(top), int(bottom), int(left), int(right))
        return ops.pad2d(x, top, bottom, left, right, mode=0)

    @staticmethod
    def backward(ctx, dy):
        t, b, l, r = ctx.pads
        return ops.pad2d(dy, -t, -b, -l, -r, mode=0), None, None, None, None


def pad2d(x, top, bottom, left, right, mode=0):
    """Zero (negative: crop) padding on mrx_pad2d in both directions; reflect padding (the one-pixel fix of odd sizes in the U-Net's up path, unet_block.py:215-222)
    recorded by torch."""
    if mode == 0 and min(top, bottom, left, right) >= 0 or mode == 0 and max(top, bottom, left, right) <= 0:
        return _ZeroPad2d.apply(x, top, bottom, left, right)
    return F.pad(x, (left, right, top, bottom), mode="constant" if mode == 0 else "reflect")


def concat_channels(a, b):
    return torch.cat([a, b], dim=1)


class _GroupNorm(torch.autograd.Function):
    """unet_block.py:71-84 with its three results (normalised tensor, mean, unbiased std): statistics + apply kernels forward, mrx_group_norm_bwd backward
    (two launches; the gradients of mean and std -- NormUnet un-normalises with them at its end -- enter the same pass)."""

    @staticmethod
    def forward(ctx, x, groups):
        out, mean, std = ops.group_norm(x, groups)
        ctx.save_for_backward(out, std)
        ctx.groups = int(groups)
        return out, mean, std

    @staticmethod
    def backward(ctx, dy, dmean, dstd):
        out, std = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(out)
        return ops.group_norm_bwd(dy, out, std, ctx.groups, dmean, dstd), None


def group_norm(x, groups):
    """unet_block.py:71-84."""
    return _GroupNorm.apply(x, groups)


class _GroupUnnorm(torch.autograd.Function):
    """unet_block.py:86-90: x * std + mean per group."""

    @staticmethod
    def forward(ctx, x, mean, std, groups):
        ctx.save_for_backward(x, std)
        ctx.groups = int(groups)
        ctx.stat_shapes = (tuple(mean.shape), tuple(std.shape))
        return ops.group_unnorm(x, mean, std, groups)

    @staticmethod
    def backward(ctx, dy):
        x, std = ctx.saved_tensors
        dx, dmean, dstd = ops.group_unnorm_bwd(dy, x, std, ctx.groups)
        return dx, dmean.reshape(ctx.stat_shapes[0]), dstd.reshape(ctx.stat_shapes[1]), None


def group_unnorm(x, mean, std, groups):
    """unet_block.py:86-90."""
    return _GroupUnnorm.apply(x, mean, std, groups)


# ---- Fourier transforms and the coil operators -------------------------------------------------------------------------------------------
def _ratio(x, normalization, spatial_dims):
    """adjoint(fft2_n) = r * ifft2_n and adjoint(ifft2_n) = fft2_n / r with r = N ("backward"), 1 ("ortho"), 1 / N ("forward")."""
    dims = spatial_dims if spatial_dims is not None else [-2, -1]
    xc = x.shape[:-1]
    n = 1
    for d in dims:
        n *= int(xc[d])
    return {"backward": float(n), "ortho": 1.0, "forward": 1.0 / n}[normalization or "backward"]


class _Fft2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, centered, normalization, spatial_dims, inverse):
        ctx.cfg = (centered, normalization, spatial_dims, inverse)
        f = fft.ifft2 if inverse else fft.fft2
        return f(x, centered=centered, normalization=normalization, spatial_dims=spatial_dims)

    @staticmethod
    def backward(ctx, dy):
        centered, normalization, spatial_dims, inverse = ctx.cfg
        r = _ratio(dy, normalization, spatial_dims)
        f = fft.fft2 if inverse else fft.ifft2
        g = f(dy.contiguous(), centered=centered, normalization=normalization, spatial_dims=spatial_dims)
        return g * (1.0 / r if inverse else r), None, None, None, None


def fft2(x, centered=False, normalization="backward", spatial_dims=None):
    return _Fft2.apply(x, centered, normalization, spatial_dims, False)


def ifft2(x, centered=False, normalization="backward", spatial_dims=None):
    return _Fft2.apply(x, centered, normalization, spatial_dims, True)


def complex_mul(a, b):
    return torch.stack([a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1], a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0]], -1)


def complex_conj(a):
    return torch.stack([a[..., 0], -a[..., 1]], -1)


class _SensExpand(torch.autograd.Function):
    """out_c = fft2(x * S_c) (vn_block.py:51-69), forward on the fused kernel; backward: G_c = adjoint-fft2(dy_c), then ONE pointwise pass for
    dx = sum_c conj(S_c) G_c and dS_c = conj(x) G_c (mrx_sens_expand_bwd_pw)."""

    @staticmethod
    def forward(ctx, x, sens, centered, normalization, spatial_dims):
        ctx.save_for_backward(x, sens)
        ctx.cfg = (centered, normalization, spatial_dims, tuple(x.shape))
        return ops.sens_expand(x, sens, centered, normalization, spatial_dims)

    @staticmethod
    def backward(ctx, dy):
        x, sens = ctx.saved_tensors
        centered, normalization, spatial_dims, xshape = ctx.cfg
        r = _ratio(dy, normalization, spatial_dims)                      # adjoint(fft2) = r * ifft2
        G = fft.ifft2(dy.contiguous(), centered=centered, normalization=normalization, spatial_dims=spatial_dims)
        dx, dS = ops.sens_expand_bwd_pointwise(G, sens, x, ctx.needs_input_grad[0], ctx.needs_input_grad[1], r)
        return (dx.reshape(xshape) if dx is not None else None), dS, None, None, None


def sens_expand(x, sens, centered, normalization, spatial_dims=None, hybrid=False):
    """vn_block.py:51-69: fft2(x * S)."""
    return _SensExpand.apply(x, sens, centered, normalization, spatial_dims)


class _SensReduce(torch.autograd.Function):
    """sum_c conj(S_c) * ifft2(k)_c (vn_block.py:71-87), forward on the fused kernel; backward: dk = adjoint-ifft2(S_c dy) = the fused sens_expand / r,
    dS_c = conj(dy) * ifft2(k)_c (one transform of the saved k-space + mrx_cmul_bcast)."""

    @staticmethod
    def forward(ctx, k, sens, centered, normalization, spatial_dims):
        ctx.save_for_backward(k, sens)
        ctx.cfg = (centered, normalization, spatial_dims)
        return ops.sens_reduce(k, sens, centered, normalization, spatial_dims)

    @staticmethod
    def backward(ctx, dy):
        k, sens = ctx.saved_tensors
        centered, normalization, spatial_dims = ctx.cfg
        dy = dy.contiguous()
        dk = dS = None
        if ctx.needs_input_grad[0]:
            r = _ratio(k, normalization, spatial_dims)                   # adjoint(ifft2) = fft2 / r
            dk = ops.sens_expand(dy, sens, centered, normalization, spatial_dims)
            if r != 1.0:
                dk.mul_(1.0 / r)
        if ctx.needs_input_grad[1]:
            img = fft.ifft2(k, centered=centered, normalization=normalization, spatial_dims=spatial_dims)
            dS = ops.cmul_bcast(img, dy, conj_v=True)
        return dk, dS, None, None, None


def sens_reduce(k, sens, centered, normalization, spatial_dims=None, work=None, hybrid=False):
    """vn_block.py:71-87 without the keepdim: sum_c conj(S_c) * ifft2(k)_c."""
    return _SensReduce.apply(k, sens, centered, normalization, spatial_dims)


class _DcCombine(torch.autograd.Function):
    """base - where(mask, pred - ref, 0) * dc_weight - eta_k (vn_block.py:113-119) on mrx_dc_combine / mrx_dc_combine_bwd.  base None: base = pred
    (the VarNet block's call) -- one input, whose whole gradient dy - where(mask, dy, 0) * w comes out of the one backward pass."""

    @staticmethod
    def forward(ctx, base, pred, ref, mask, dc_weight, eta_k):
        ctx.save_for_backward(pred, ref, mask, dc_weight)
        ctx.same = base is None
        return ops.dc_combine(pred if base is None else base, pred, ref, mask, dc_weight, eta_k)

    @staticmethod
    def backward(ctx, dy):
        pred, ref, mask, dc_weight = ctx.saved_tensors
        dy = dy.contiguous()
        nb, npd, _, _, nw, ne = ctx.needs_input_grad
        dpred, deta, dw = ops.dc_combine_bwd(dy, pred, ref, mask, dc_weight, npd, ne, nw, ctx.same)
        return (dy if (nb and not ctx.same) else None), dpred, None, None, (dw.reshape(dc_weight.shape) if dw is not None else None), deta


def dc_combine(base, pred, ref, mask, dc_weight, eta_k):
    """vn_block.py:113-119: base - where(mask, pred - ref, 0) * dc_weight - eta_k."""
    return _DcCombine.apply(None if base is pred else base, pred, ref, mask, dc_weight, eta_k)


def coil_combination(data, sens, method="SENSE", dim=1):
    """utils.py:251-272."""
    if method == "SENSE":
        return complex_mul(data, complex_conj(sens)).sum(dim)
    if method == "RSS":
        return torch.sqrt((data ** 2).sum(dim))
    raise ValueError("Output type not supported.")


# ---- gated recurrent cells (rnn_cells.py:112-127, 249-261): convolutions AND gates on the HIP kernels in both directions ------------------------------
class _Gates(torch.autograd.Function):
    """h' = gates(ih, hh, hx): mrx_gru_gates / mrx_mgu_gates forward, mrx_gru_gates_bwd / mrx_mgu_gates_bwd backward (one launch each; the gates are recomputed
    from the saved ih / hh -- what torch would have saved as ~ten intermediate planes)."""

    @staticmethod
    def forward(ctx, ih, hh, hx, gru):
        ctx.save_for_backward(ih, hh, hx)
        ctx.gru = bool(gru)
        return (ops.gru_gates if gru else ops.mgu_gates)(ih, hh, hx)

    @staticmethod
    def backward(ctx, dy):
        ih, hh, hx = ctx.saved_tensors
        dih, dhh, dh = (ops.gru_gates_bwd if ctx.gru else ops.mgu_gates_bwd)(dy, ih, hh, hx)
        return dih, dhh, dh, None


def gru_gates(ih, hh, hx):
    return _Gates.apply(ih, hh, hx, True)


def mgu_gates(ih, hh, hx):
    return _Gates.apply(ih, hh, hx, False)
