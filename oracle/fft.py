"""Oracle: FFT operators (reference mridc/collections/common/parts/fft.py).  Test infrastructure."""
import numpy as np
import torch


def _norm(normalization):
    # fft.py:80 / :158 -- "none" means torch's default (= backward)
    return None if str(normalization).lower() == "none" else normalization


def roll_one_dim(data, shift, dim):
    """fft.py:169-202.  out[i] = data[(i - shift) mod n] along `dim`."""
    n = data.size(dim)
    shift = int(shift) % n
    if shift == 0:
        return data
    idx = (torch.arange(n) - shift) % n
    return data.index_select(dim, idx)


def roll(data, shift, dim):
    """fft.py:205-240."""
    if len(shift) != len(dim):
        raise ValueError("len(shift) must match len(dim)")
    for s, d in zip(shift, dim):
        data = roll_one_dim(data, s, d)
    return data


def fftshift(data, dim=None):
    """fft.py:243-281: shift by n//2 on every listed dim (all dims when None)."""
    if dim is None:
        dim = list(range(data.dim()))
    return roll(data, [data.shape[d] // 2 for d in dim], list(dim))


def ifftshift(data, dim=None):
    """fft.py:284-322: shift by (n+1)//2."""
    if dim is None:
        dim = list(range(data.dim()))
    return roll(data, [(data.shape[d] + 1) // 2 for d in dim], list(dim))


def _xform(data, centered, normalization, spatial_dims, inverse):
    # fft.py:66-88 (fft2) and :144-166 (ifft2): real view -> complex, default dims applied on the
    # complex view, optional ifftshift before / fftshift after, always returns the real view.
    if data.shape[-1] == 2:
        data = torch.view_as_complex(data.contiguous())
    dims = [-2, -1] if spatial_dims is None else list(spatial_dims)
    if centered:
        data = ifftshift(data, dim=dims)
    fn = torch.fft.ifft2 if inverse else torch.fft.fft2
    data = fn(data, dim=dims, norm=_norm(normalization))
    if centered:
        data = fftshift(data, dim=dims)
    return torch.view_as_real(data)


def fft2(data, centered=False, normalization="backward", spatial_dims=None):
    """fft.py:13-88."""
    return _xform(data, centered, normalization, spatial_dims, inverse=False)


def ifft2(data, centered=False, normalization="backward", spatial_dims=None):
    """fft.py:91-166."""
    return _xform(data, centered, normalization, spatial_dims, inverse=True)


# ---------------------------------------------------------------------------------------------
# Independent float64 restatement (textbook DFT via numpy), used to pin the torch path and as the
# "truth" when comparing fp32 implementations with a norm-relative criterion.
# ---------------------------------------------------------------------------------------------
def fft2_np64(x, centered=False, normalization="backward", inverse=False):
    """x: numpy complex array, transform over the last two axes, float64 arithmetic."""
    x = np.asarray(x).astype(np.complex128)
    if centered:
        x = np.fft.ifftshift(x, axes=(-2, -1))
    nrm = _norm(normalization) or "backward"
    x = (np.fft.ifft2 if inverse else np.fft.fft2)(x, axes=(-2, -1), norm=nrm)
    if centered:
        x = np.fft.fftshift(x, axes=(-2, -1))
    return x


def dft_matrix(n, inverse=False):
    """Plain DFT matrix in float64 -- a transform restated from the definition, for small n."""
    k = np.arange(n)
    s = 1.0 if inverse else -1.0
    return np.exp(s * 2j * np.pi * np.outer(k, k) / n)


def fft2_definition(x, normalization="backward", inverse=False):
    """2-D DFT from the definition (two matrix products), float64.  Small sizes only."""
    x = np.asarray(x).astype(np.complex128)
    h, w = x.shape[-2:]
    out = dft_matrix(h, inverse) @ x @ dft_matrix(w, inverse).T
    nrm = _norm(normalization) or "backward"
    n = h * w
    if nrm == "ortho":
        out = out / np.sqrt(n)
    elif (nrm == "forward" and not inverse) or (nrm == "backward" and inverse):
        out = out / n
    return out
