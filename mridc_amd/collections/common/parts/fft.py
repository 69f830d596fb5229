"""FFT operators -- drop-in for `mridc.collections.common.parts.fft` (reference fft.py:10), HIP backed.

fft2 / ifft2 run as an LDS-resident mixed-radix row pass + column pass (csrc/fft.hip); centred shifts are index
math inside the kernels, never materialised rolls.  roll / fftshift / ifftshift are one bit-exact gather kernel.
"""
from typing import List, Sequence, Union

import numpy as np
import torch

from mridc_amd import _lib

__all__ = ["fft2", "ifft2", "fftshift", "ifftshift", "roll", "roll_one_dim", "fft2c", "ifft2c"]


def _norm_code(normalization: str) -> int:
    key = str(normalization).lower()
    if key not in _lib.NORM:
        raise ValueError(f"Unknown fft normalization '{normalization}' (expected backward, ortho, forward or none)")
    return _lib.NORM[key]


def _xform(data: torch.Tensor, centered: bool, normalization: str, spatial_dims, inverse: bool) -> torch.Tensor:
    _lib.require_gpu(data)
    if data.shape[-1] == 2 and not data.is_complex():       # fft.py:66-67
        real = data
    elif data.is_complex():                                  # appendix D.18: already-complex input
        real = torch.view_as_real(data)
    else:
        raise ValueError("fft2/ifft2 expect a real view [...,2] or a complex tensor")
    real = _lib.f32c(real)
    nd = real.dim() - 1                                      # dims of the complex view
    dims = [-2, -1] if spatial_dims is None else list(spatial_dims)   # fft.py:69-72 (applied on the complex view)
    if len(dims) != 2:
        raise ValueError("spatial_dims must name exactly two dimensions")
    dims = [d % nd for d in dims]
    if dims[0] == dims[1]:
        raise ValueError("spatial_dims must be distinct")
    norm = _norm_code(normalization)
    perm = None
    if sorted(dims) != [nd - 2, nd - 1]:
        rest = [d for d in range(nd) if d not in dims]
        perm = rest + sorted(dims) + [nd]
        real = real.permute(perm).contiguous()
    H, W = int(real.shape[-3]), int(real.shape[-2])
    batch = int(real.numel() // (2 * H * W)) if H * W > 0 else 0
    L = _lib.lib()
    if batch > 0 and max(H, W) > int(L.mrx_fft_max_len()):
        out = _xform_long(real, H, W, centered, str(normalization).lower(), inverse)      # beyond the LDS-resident transform length
        batch = 0
    else:
        out = torch.empty_like(real)
    if batch > 0:
        _lib.check(L.mrx_fft2(_lib.ptr(real), _lib.ptr(out), batch, H, W, int(inverse), norm, int(bool(centered)),
                              _lib.stream_ptr()), "mrx_fft2")
    if perm is not None:
        inv = [0] * len(perm)
        for i, p in enumerate(perm):
            inv[p] = i
        out = out.permute(inv).contiguous()
    return out


# ---- lengths beyond the LDS-resident limit (mrx_fft_max_len, 4096): the four-step algorithm over the existing kernels -----------------------
# torch.fft in the reference has no length limit (fft.py:77-81,155-159).  N = N1 * N2 (both within the limit), n = N2 n1 + n2, k = k1 + N1 k2:
#   X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) * [ W_N^(n2 k1) * sum_n1 x[N2 n1 + n2] W_N1^(n1 k1) ]
# = column transforms of length N1 (mrx_fft_cols), a twiddle multiplication (mrx_complex_mul, table computed in float64), row transforms
# of length N2 (mrx_fft2 on one-row images) and a transpose.  An edge path: composed from launches, not fused.
_TWIDDLES = {}


def _split_length(n: int, limit: int):
    best = None
    for n1 in range(min(limit, n), 0, -1):
        if n % n1 == 0 and n // n1 <= limit:
            if best is None or abs(n1 - n // n1) < abs(best - n // best):
                best = n1
    if best is None:
        raise NotImplementedError(f"fft length {n}: no factorisation into two lengths <= {limit} (prime lengths beyond the LDS-resident "
                                  "limit are not supported)")
    return best, n // best


def _fft_last_axis(x: torch.Tensor, inverse: bool) -> torch.Tensor:
    """Unnormalised forward / (1/N)-scaled inverse transform along the last complex axis of x [M, N, 2]."""
    from mridc_amd.collections.common.parts import utils
    L = _lib.lib()
    M, N = int(x.shape[0]), int(x.shape[1])
    limit = int(L.mrx_fft_max_len())
    out = torch.empty_like(x)
    if N <= limit:
        _lib.check(L.mrx_fft2(_lib.ptr(x), _lib.ptr(out), M, 1, N, int(inverse), _lib.NORM["backward"], 0, _lib.stream_ptr()), "mrx_fft2")
        return out
    n1, n2 = _split_length(N, limit)
    _lib.check(L.mrx_fft_cols(_lib.ptr(x), _lib.ptr(out), M, n1, n2, int(inverse), _lib.NORM["backward"], 0, _lib.stream_ptr()), "mrx_fft_cols")
    key = (n1, n2, bool(inverse), str(x.device))
    tw = _TWIDDLES.get(key)
    if tw is None:
        ang = (2.0 * np.pi / N) * np.outer(np.arange(n1, dtype=np.float64), np.arange(n2, dtype=np.float64))
        ang = ang if inverse else -ang
        tw = torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)).to(x.device)
        if len(_TWIDDLES) >= 8:
            _TWIDDLES.pop(next(iter(_TWIDDLES)))
        _TWIDDLES[key] = tw
    y = utils.complex_mul(out.view(M, n1, n2, 2), tw.unsqueeze(0))
    z = torch.empty_like(y)
    _lib.check(L.mrx_fft2(_lib.ptr(y), _lib.ptr(z), M * n1, 1, n2, int(inverse), _lib.NORM["backward"], 0, _lib.stream_ptr()), "mrx_fft2")
    return z.transpose(1, 2).reshape(M, N, 2).contiguous()          # k = k1 + n1 * k2


def _xform_long(real: torch.Tensor, H: int, W: int, centered: bool, normalization: str, inverse: bool) -> torch.Tensor:
    shape = real.shape
    x = real.reshape(-1, H, W, 2)
    if centered:                                                   # fft.py:74-75: ifftshift before, fftshift after
        x = _roll_many(x, [(H + 1) // 2, (W + 1) // 2], [1, 2])
    x = _fft_last_axis(x.reshape(-1, W, 2), inverse).view(-1, H, W, 2)
    x = x.transpose(1, 2).contiguous()                             # [B, W, H, 2]: the H axis last
    x = _fft_last_axis(x.reshape(-1, H, 2), inverse).view(-1, W, H, 2).transpose(1, 2).contiguous()
    n = float(H) * float(W)
    scale = {"backward": 1.0, "none": 1.0, "ortho": (n ** 0.5 if inverse else n ** -0.5), "forward": (n if inverse else 1.0 / n)}[normalization]
    if scale != 1.0:
        y = torch.empty_like(x)
        _lib.check(_lib.lib().mrx_scale(_lib.ptr(x), _lib.ptr(y), x.numel(), float(scale), 0, _lib.stream_ptr()), "mrx_scale")
        x = y
    if centered:
        x = _roll_many(x, [H // 2, W // 2], [1, 2])
    return x.reshape(shape)


def fft2(data: torch.Tensor, centered: bool = False, normalization: str = "backward",
         spatial_dims: Sequence[int] = None) -> torch.Tensor:
    """Reference fft.py:13-88.  Returns the real view [...,2]."""
    return _xform(data, centered, normalization, spatial_dims, inverse=False)


def ifft2(data: torch.Tensor, centered: bool = False, normalization: str = "backward",
          spatial_dims: Sequence[int] = None) -> torch.Tensor:
    """Reference fft.py:91-166."""
    return _xform(data, centered, normalization, spatial_dims, inverse=True)


def fft2c(data, normalization="ortho", spatial_dims=None):
    """fastMRI/ATOMMIC-style alias named by BASELINE.json: centred fft2."""
    return fft2(data, centered=True, normalization=normalization, spatial_dims=spatial_dims)


def ifft2c(data, normalization="ortho", spatial_dims=None):
    return ifft2(data, centered=True, normalization=normalization, spatial_dims=spatial_dims)


def _roll_many(data: torch.Tensor, shifts: List[int], dims: List[int]) -> torch.Tensor:
    _lib.require_gpu(data)
    x = data.contiguous()
    nd = x.dim()
    total = [0] * nd
    for s, d in zip(shifts, dims):                           # successive rolls compose additively per dim
        total[d % nd] += int(s)
    if x.numel() == 0 or all(t % n == 0 for t, n in zip(total, x.shape)):
        return data                                          # fft.py:196-197 returns the input itself
    if nd > 8:
        raise ValueError("roll supports at most 8 dimensions")
    out = torch.empty_like(x)
    L = _lib.lib()
    _lib.check(L.mrx_roll(_lib.ptr(x), _lib.ptr(out), x.element_size(), nd, _lib.i64_array(x.shape),
                          _lib.i64_array(total), _lib.stream_ptr()), "mrx_roll")
    return out


def roll_one_dim(data: torch.Tensor, shift: int, dim: int) -> torch.Tensor:
    """Reference fft.py:169-202."""
    return _roll_many(data, [shift], [dim])


def roll(data: torch.Tensor, shift: List[int], dim: Union[List[int], Sequence[int]]) -> torch.Tensor:
    """Reference fft.py:205-240."""
    if len(shift) != len(dim):
        raise ValueError("len(shift) must match len(dim)")
    return _roll_many(data, list(shift), list(dim))


def fftshift(data: torch.Tensor, dim: Union[List[int], Sequence[int]] = None) -> torch.Tensor:
    """Reference fft.py:243-281: shift n//2."""
    dim = list(range(data.dim())) if dim is None else list(dim)
    return roll(data, [int(np.floor_divide(data.shape[d], 2)) for d in dim], dim)


def ifftshift(data: torch.Tensor, dim: Union[List[int], Sequence[int]] = None) -> torch.Tensor:
    """Reference fft.py:284-322: shift (n+1)//2."""
    dim = list(range(data.dim())) if dim is None else list(dim)
    return roll(data, [int(np.floor_divide(data.shape[d] + 1, 2)) for d in dim], dim)
