"""Drop-in for `mridc.collections.reconstruction.models.sigmanet.sensitivity_net` (reference sensitivity_net.py:9-310): the complex
instance normalisation around a regulariser and the unrolled network x <- D(x - R(x)).

The normalisation whitens the (re, im) pairs of an image stack: with m the mean over every real and imaginary entry and C the 2x2
covariance of (re - m, im - m) per batch element, `normalize` is x -> C^(-1/2) (x - m) clamped to [-6, 6] and `unnormalize` its inverse
y -> C^(1/2) y + m.  Here the statistics are three launches of libmridc_amd (`mrx_cnorm_stats`: fixed-order double-precision reductions, the
2x2 matrix square root and its inverse formed on the device) and each direction is ONE pass that also does the reference's
view / permute between [B,C,H,W,2] and the regulariser's [B C, 2, H, W] (`mrx_cnorm_apply`, `mrx_cnorm_unapply`) -- no host
synchronisation, no torch arithmetic.

Shapes are the reference's, quirks included: the covariance sums run over every axis between the batch axis and the complex axis but are
divided by `shape[2] * shape[3] - 1` of whatever rank arrives (H W - 1 for [B,C,H,W,2]; 2 W - 1 for a 4-D [B,H,W,2] image); a 4-D image
leaves the wrapper as [B,B,H,W,2] because its [B,1,1,1] coefficients broadcast against [B,H,W] (B = 1 in the reference's own use:
[1,1,H,W,2]).  One difference: the reference takes C^(1/2) from an eigen-decomposition that is 0 / 0 when c_xy = 0 exactly; the kernel
forms the principal square root directly (csrc/cnorm.hip), the same matrix wherever the reference's is defined."""
import numpy as np
import torch

from mridc_amd import _lib


def matrix_invert(xx, xy, yx, yy):
    """sensitivity_net.py:9-12: the inverse of [[xx, xy], [yx, yy]], entry by entry."""
    det = xx * yy - xy * yx
    return yy / det, -(xy / det), -(yx / det), xx / det


def _as_stack(x):
    """(x5 [B,C,H,W,2] contiguous fp32, divisor, squeeze-back shape): the 5-D stack whose image (b, c) gets batch element b's coefficients,
    the reference's covariance divisor and the shape the result has in the reference."""
    if x.shape[-1] != 2:
        raise AssertionError
    x = _lib.f32c(x)
    divisor = int(x.shape[2]) * int(x.shape[3]) - 1
    if x.dim() == 5:
        return x, divisor, tuple(x.shape)
    if x.dim() == 4:                                  # [B,H,W,2]: coefficients [B,1,1,1] against [B,H,W] -> [B,B,H,W,2]
        B = int(x.shape[0])
        stack = x.unsqueeze(0).expand(B, *x.shape).contiguous() if B > 1 else x.unsqueeze(0)
        return stack, divisor, (B, *x.shape)
    raise NotImplementedError(f"ComplexInstanceNorm: rank-{x.dim()} input")


class ComplexInstanceNorm(torch.nn.Module):
    """sensitivity_net.py:15-118.  `set_normalization(x)` measures x; `normalize` / `unnormalize` apply what was measured.  The measured
    quantities are kept as one device tensor `coef` [B,9] = (m, C^(1/2) row-major, C^(-1/2) row-major) and exposed under the reference's
    attribute names (`mean`, `cov_xx_half`, ... as [B,1,1,1] views)."""

    def __init__(self):
        super().__init__()
        self.coef = None
        self.mean = 0
        self.cov_xx_half = self.cov_yy_half = 1 / np.sqrt(2)
        self.cov_xy_half = self.cov_yx_half = 0

    # ---- measuring ---------------------------------------------------------------------------------------------------------------------
    def _measure(self, x, center):
        """coef [B,9] of the batch elements of x (statistics over x's own entries; `center` False: the data is taken as mean-free)."""
        if x.dim() not in (4, 5) or x.shape[-1] != 2:
            raise AssertionError
        x = _lib.f32c(x)
        _lib.require_gpu(x)
        B = int(x.shape[0])
        per_b = x[0].numel() // 2
        L = _lib.lib()
        coef = torch.empty(B, 9, dtype=torch.float32, device=x.device)
        work = torch.empty(int(L.mrx_cnorm_work_doubles(B)), dtype=torch.float64, device=x.device)
        _lib.check(L.mrx_cnorm_stats(_lib.ptr(x), B, per_b, float(int(x.shape[2]) * int(x.shape[3]) - 1), int(bool(center)), _lib.ptr(coef),
                                     _lib.ptr(work), _lib.stream_ptr()), "mrx_cnorm_stats")
        return coef

    def _publish(self, coef, mean=None):
        self.coef = coef
        cols = [coef[:, i].reshape(-1, 1, 1, 1) for i in range(5)]
        self.mean = cols[0][:1] if mean is None else mean          # the reference's mean is one scalar, kept as [1,1,1,1]
        self.cov_xx_half, self.cov_xy_half, self.cov_yx_half, self.cov_yy_half = cols[1:]

    def set_normalization(self, input):
        """sensitivity_net.py:85-93: the mean over all entries of `input`, the covariance of what is left."""
        self._publish(self._measure(input, center=True))

    def complex_pseudocovariance(self, data):
        """sensitivity_net.py:34-83: the square root of the covariance of mean-free `data` (the mean is left as it is)."""
        coef = self._measure(data, center=False)
        keep = self.mean
        self._publish(coef, mean=keep)

    def complex_instance_norm(self, x, eps=1e-5):
        """sensitivity_net.py:26-32: per-sample complex mean of the coil sum, then the covariance of the centred stack."""
        mean = x.sum(dim=1, keepdim=True).mean(dim=(1, 2, 3), keepdim=True)
        self.mean = mean
        self.complex_pseudocovariance(x - mean)

    # ---- applying ------------------------------------------------------------------------------------------------------------------------
    def _coef_for(self, x):
        if self.coef is None:
            raise RuntimeError("ComplexInstanceNorm: set_normalization() has not been called")
        if not torch.is_tensor(self.mean) or self.mean.numel() != 1:
            raise NotImplementedError("ComplexInstanceNorm: normalize / unnormalize after complex_instance_norm (a per-sample complex mean)")
        return self.coef

    def _channel_first(self, x):
        """clamp(C^(-1/2) (x - m), -6, 6) laid out [B C, 2, H, W], and the reference shape of the normalised stack."""
        coef = self._coef_for(x)
        x5, _, shape = _as_stack(x)
        B, C, H, W = (int(v) for v in x5.shape[:4])
        if B != coef.shape[0]:
            raise RuntimeError(f"ComplexInstanceNorm: measured {coef.shape[0]} batch elements, asked to normalise {B}")
        out = torch.empty(B * C, 2, H, W, dtype=torch.float32, device=x5.device)
        _lib.check(_lib.lib().mrx_cnorm_apply(_lib.ptr(x5), _lib.ptr(coef), _lib.ptr(out), B, C, H * W, _lib.stream_ptr()), "mrx_cnorm_apply")
        return out, shape

    def _from_channel_first(self, y, shape):
        coef = self._coef_for(y)
        y = _lib.f32c(y)
        B, C, H, W = (int(v) for v in shape[:4])
        out = torch.empty(B, C, H, W, 2, dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().mrx_cnorm_unapply(_lib.ptr(y), _lib.ptr(coef), _lib.ptr(out), B, C, H * W, _lib.stream_ptr()), "mrx_cnorm_unapply")
        return out

    def normalize(self, x):
        """sensitivity_net.py:95-108 ([...,2] in, [...,2] out)."""
        y, shape = self._channel_first(x)
        B, C, H, W = shape[:4]
        return y.view(B, C, 2, H, W).permute(0, 1, 3, 4, 2).contiguous()

    def unnormalize(self, x):
        """sensitivity_net.py:110-117 ([B,C,H,W,2] in and out)."""
        x = _lib.f32c(x)
        if x.dim() != 5:
            raise NotImplementedError("ComplexInstanceNorm.unnormalize: [B,C,H,W,2] input")
        B, C, H, W = (int(v) for v in x.shape[:4])
        return self._from_channel_first(x.permute(0, 1, 4, 2, 3).reshape(B * C, 2, H, W), x.shape)

    def forward(self, input):
        return self.normalize(input)


class ComplexNormWrapper(torch.nn.Module):
    """sensitivity_net.py:121-139: measure, normalise, run `model` on [B C, 2, H, W], un-normalise -- the two layout changes ride inside the
    normalisation passes."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.complex_instance_norm = ComplexInstanceNorm()

    def forward(self, input):
        norm = self.complex_instance_norm
        from mridc_amd import diff
        if diff.active(input, *self.model.parameters(), training=self.training):
            return self._forward_recorded(input)           # the kernels below write raw buffers: their outputs carry no grad_fn
        # (eval() without torch.no_grad() and without an input gradient keeps the fused kernels: the rule of diff.active)
        norm.set_normalization(input)
        channel_first, shape = norm._channel_first(input)
        return norm._from_channel_first(self.model(channel_first), shape)

    def _forward_recorded(self, input):
        """The same map in torch device arithmetic, recorded by autograd (gradients flow through the regulariser AND through the measured covariance,
        as in the reference, whose mean is a detached scalar: sensitivity_net.py:87): C^(1/2) in the closed form of csrc/cnorm.hip,
        (C + sqrt(det C) I) / sqrt(tr C + 2 sqrt(det C)).  Used whenever gradients are being recorded -- the stage-wise training helpers of
        SensitivityNetwork need it; inference (no_grad) takes the kernels."""
        if input.dim() != 5 or input.shape[-1] != 2:
            raise NotImplementedError("ComplexNormWrapper with gradients: [B,C,H,W,2] input")
        x = input.float()
        B, C, H, W = (int(v) for v in x.shape[:4])
        mean = x.mean().detach()
        re, im = (x - mean).unbind(-1)
        n1 = float(H * W - 1)                                # the reference's divisor: shape[2] * shape[3] - 1
        cxx, cyy, cxy = ((re * re).sum((1, 2, 3), keepdim=True) / n1, (im * im).sum((1, 2, 3), keepdim=True) / n1,
                         (re * im).sum((1, 2, 3), keepdim=True) / n1)
        s_ = torch.sqrt(torch.clamp(cxx * cyy - cxy * cxy, min=0.0))
        t_ = torch.sqrt(cxx + cyy + 2.0 * s_)
        hxx, hxy, hyy = (cxx + s_) / t_, cxy / t_, (cyy + s_) / t_
        ixx, ixy, iyx, iyy = matrix_invert(hxx, hxy, hxy, hyy)
        z = torch.stack([ixx * re + ixy * im, iyx * re + iyy * im], dim=-1).clamp(-6, 6)
        y = self.model(z.reshape(B * C, H, W, 2).permute(0, 3, 1, 2))
        y = y.permute(0, 2, 3, 1).reshape(B, C, H, W, 2)
        yr, yi = y.unbind(-1)
        norm = self.complex_instance_norm
        norm.mean, norm.cov_xx_half, norm.cov_xy_half, norm.cov_yx_half, norm.cov_yy_half = mean.reshape(1, 1, 1, 1), hxx, hxy, hxy, hyy
        return torch.stack([hxx * yr + hxy * yi, hxy * yr + hyy * yi], dim=-1) + mean


class SensitivityNetwork(torch.nn.Module):
    """sensitivity_net.py:142-310: `num_iter` rounds of x <- gradD(x - gradR(x), y, smaps, mask).  With `shared_params` one regulariser /
    data layer pair serves every round.  The stage-wise training helpers keep the reference's book-keeping (`is_trainable`); like the
    reference they mark parameters through an attribute named `require_grad_`, which autograd does not read."""

    def __init__(self, num_iter, model, datalayer, shared_params=True, save_space=False, reset_cache=False):
        super().__init__()
        self.shared_params = shared_params
        self.num_iter_total = num_iter
        self.num_iter = 1 if shared_params else num_iter
        self.is_trainable = [True] * num_iter
        self.gradR = torch.nn.ModuleList(ComplexNormWrapper(model) for _ in range(self.num_iter))
        self.gradD = torch.nn.ModuleList(datalayer for _ in range(self.num_iter))
        self.save_space = save_space
        self.reset_cache = reset_cache

    def _rounds(self):
        """How many rounds run: all of them with shared parameters, otherwise up to the last stage that is being trained."""
        if self.shared_params:
            return self.num_iter_total
        last_trained = max(i for i, flag in enumerate(self.is_trainable) if flag)
        return min(last_trained + 1, self.num_iter)

    def forward(self, x, y, smaps, mask):
        for i in range(self._rounds()):
            stage = i % self.num_iter
            x = self.gradD[stage](x - self.gradR[stage](x), y, smaps, mask)
        return x

    forward_save_space = forward                       # the reference's memory-saving variant computes the same values

    # ---- stage-wise training book-keeping (sensitivity_net.py:214-310) --------------------------------------------------------------------
    def _mark(self, i, flag):
        for p in self.gradR[i].parameters():
            p.require_grad_ = flag
        self.is_trainable[i] = flag

    def freeze(self, i):
        self._mark(i, False)

    def unfreeze(self, i):
        self._mark(i, True)

    def freeze_all(self):
        for i in range(self.num_iter):
            self._mark(i, False)

    def unfreeze_all(self):
        for i in range(self.num_iter):
            self._mark(i, True)

    def copy_params(self, src_i, trg_j):
        with torch.no_grad():
            for dst, src in zip(self.gradR[trg_j].parameters(), self.gradR[src_i].parameters()):
                dst.copy_(src)

    def stage_training_init(self):
        self.freeze_all()
        self.unfreeze(0)

    def stage_training_transition_i(self, copy=False):
        """Move the trainable stage one step on (optionally seeding it with its predecessor's weights); after the last stage, train all."""
        if self.shared_params or all(self.is_trainable):
            return
        current = next((i for i in range(self.num_iter) if self.is_trainable[i]), self.num_iter - 1)
        if current == self.num_iter - 1:
            self.unfreeze_all()
            return
        self.freeze(current)
        self.unfreeze(current + 1)
        if copy:
            self.copy_params(current, current + 1)
