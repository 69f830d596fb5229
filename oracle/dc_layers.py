"""Oracle: the sigmanet data-consistency layers (reference models/sigmanet/dc_layers.py), forward passes.  Test infrastructure.

The reference indexes the coil axis as -4 and the batch axis as -5 on 5-D tensors, so with a 4-D image `x` [B,H,W,2] the
gradient-descent and variable-splitting layers sum over the *batch* axis of the coil images and return [C,H,W,2]; that is
restated as is.  The CG layer runs only with a 5-D image [B,1,H,W,2] (its solver reshapes alpha to 5-D)."""
import torch

from . import fft as offt
from . import utils as outils


def _A(x, smaps, mask, c, n, sd):
    return (offt.fft2(outils.complex_mul(x.expand_as(smaps), smaps), c, n, sd) * mask).sum(-4, keepdim=True)


def _AT(x, smaps, mask, c, n, sd):
    return outils.complex_mul(offt.ifft2(x * mask, c, n, sd), outils.complex_conj(smaps)).sum(-5)


def data_gd(x, y, smaps, mask, data_weight, fft_centered=True, fft_normalization="ortho", spatial_dims=None):
    """dc_layers.py:54-96: x - lambda * A^H(M(Ax) - y)."""
    A_x_y = (offt.fft2(outils.complex_mul(x.unsqueeze(-5).expand_as(smaps), smaps), fft_centered, fft_normalization, spatial_dims)
             * mask).sum(-4, keepdim=True) - y
    gradD_x = outils.complex_mul(offt.ifft2(A_x_y * mask, fft_centered, fft_normalization, spatial_dims),
                                 outils.complex_conj(smaps)).sum(-5)
    return x - data_weight * gradD_x


def data_vs(x, y, smaps, mask, alpha, beta, fft_centered=True, fft_normalization="ortho", spatial_dims=None):
    """dc_layers.py:368-402."""
    A_x = offt.fft2(outils.complex_mul(x.unsqueeze(-5).expand_as(smaps), smaps), fft_centered, fft_normalization,
                    spatial_dims).sum(-4, keepdim=True)
    k_dc = (1 - mask) * A_x + mask * (alpha * A_x + (1 - alpha) * y)
    x_dc = outils.complex_mul(offt.ifft2(k_dc, fft_centered, fft_normalization, spatial_dims), outils.complex_conj(smaps)).sum(-5)
    return beta * x + (1 - beta) * x_dc


def dc_single(x, y, mask, lambda_, fft_centered=True, fft_normalization="ortho", spatial_dims=None):
    """dc_layers.py:448-467 (single coil)."""
    A_x = offt.fft2(x, fft_centered, fft_normalization, spatial_dims)
    k_dc = (1 - mask) * A_x + mask * (lambda_ * A_x + (1 - lambda_) * y)
    return offt.ifft2(k_dc, fft_centered, fft_normalization, spatial_dims)


def _complex_dot(a, b):
    """dc_layers.py:160-165."""
    nB = a.shape[0]
    m = outils.complex_mul(a, outils.complex_conj(b))
    return torch.stack([m[..., 0].reshape(nB, -1).sum(-1), m[..., 1].reshape(nB, -1).sum(-1)], -1)


def cg_solve(x0, M, tol, max_iter):
    """dc_layers.py:167-196."""
    nB = x0.shape[0]
    x = torch.zeros_like(x0)
    r, p = x0.clone(), x0.clone()
    x0x0 = x0.pow(2).reshape(nB, -1).sum(-1)
    rr = torch.stack([r.pow(2).reshape(nB, -1).sum(-1), torch.zeros(nB)], -1)
    it = 0
    while torch.min(rr[..., 0] / x0x0) > tol and it < max_iter:
        it += 1
        q = M(p)
        d2 = _complex_dot(p, q)
        re1, im1 = rr[..., 0], rr[..., 1]
        re2, im2 = d2[..., 0], d2[..., 1]
        alpha = torch.stack([re1 * re2 + im1 * im2, im1 * re2 - re1 * im2], -1) / outils.complex_abs(d2) ** 2
        x = x + outils.complex_mul(alpha.reshape(nB, 1, 1, 1, -1), p)
        r = r - outils.complex_mul(alpha.reshape(nB, 1, 1, 1, -1), q)
        rr_new = torch.stack([r.pow(2).reshape(nB, -1).sum(-1), torch.zeros(nB)], -1)
        beta = torch.stack([rr_new[..., 0] / rr[..., 0], torch.zeros(nB)], -1)
        p = r + outils.complex_mul(beta.reshape(nB, 1, 1, 1, -1), p)
        rr = rr_new
    return x


def data_prox_cg(z, y, smaps, mask, lambdaa, tol=1e-6, max_iter=10, fft_centered=True, fft_normalization="ortho", spatial_dims=None):
    """dc_layers.py:198-257 (ConjugateGradient.forward): solve (lambda A^H A + I) x = lambda A^H y + z."""
    c, n, sd = fft_centered, fft_normalization, spatial_dims

    def M(p):
        return lambdaa * _AT(_A(p, smaps, mask, c, n, sd), smaps, mask, c, n, sd) + p

    x0 = lambdaa * _AT(y, smaps, mask, c, n, sd) + z
    return cg_solve(x0, M, tol, max_iter)
