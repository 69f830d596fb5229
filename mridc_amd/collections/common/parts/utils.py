"""Complex / coil operators -- drop-in for `mridc.collections.common.parts.utils` (reference utils.py:13-33), HIP backed."""
from typing import Any, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from mridc_amd import _lib

__all__ = [
    "is_none", "to_tensor", "tensor_to_complex_np", "complex_mul", "complex_conj", "complex_abs", "complex_abs_sq",
    "rss", "rss_complex", "sense", "coil_combination", "check_stacked_complex", "apply_mask", "mask_center",
    "batched_mask_center", "center_crop", "complex_center_crop", "center_crop_to_smallest",
]


def is_none(x: Union[Any, None]) -> bool:
    """utils.py:38-50."""
    return x is None or str(x).lower() == "none"


def to_tensor(data: np.ndarray) -> torch.Tensor:
    """utils.py:53-71 (host-side conversion)."""
    if np.iscomplexobj(data):
        data = np.stack((data.real, data.imag), axis=-1)
    return torch.from_numpy(data)


def tensor_to_complex_np(data: torch.Tensor) -> np.ndarray:
    """utils.py:74-88."""
    data = data.detach().cpu().numpy()
    return data[..., 0] + 1j * data[..., 1]


def _prod(xs):
    r = 1
    for v in xs:
        r *= int(v)
    return r


def complex_mul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """utils.py:96-118, with torch broadcasting of the leading dims."""
    if not x.shape[-1] == y.shape[-1] == 2:
        raise ValueError("Tensors do not have separate complex dim.")
    x, y = _lib.f32c(x), _lib.f32c(y)
    shape = torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    xe, ye = x.expand(*shape, 2), y.expand(*shape, 2)
    # collapse to <= 6 dims is not needed for the shapes of this path; reject deeper broadcasts explicitly
    if len(shape) > 6:
        raise ValueError("complex_mul supports at most 6 leading dimensions")
    out = torch.empty(*shape, 2, dtype=torch.float32, device=x.device)
    if len(shape) == 0:
        shape_l, xs, ys = [1], [0], [0]
    else:
        shape_l = list(shape)
        xs = [xe.stride(i) // 2 for i in range(len(shape))]
        ys = [ye.stride(i) // 2 for i in range(len(shape))]
    L = _lib.lib()
    _lib.check(L.mrx_complex_mul(_lib.ptr(x), _lib.ptr(y), _lib.ptr(out), len(shape_l), _lib.i64_array(shape_l),
                                 _lib.i64_array(xs), _lib.i64_array(ys), 0, _lib.stream_ptr()), "mrx_complex_mul")
    return out


def complex_conj(x: torch.Tensor) -> torch.Tensor:
    """utils.py:121-139."""
    if x.shape[-1] != 2:
        raise ValueError("Tensor does not have separate complex dim.")
    x = _lib.f32c(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_complex_conj(_lib.ptr(x), _lib.ptr(out), x.numel() // 2, _lib.stream_ptr()), "mrx_complex_conj")
    return out


def _abs(data, squared):
    if data.shape[-1] != 2:
        raise ValueError("Tensor does not have separate complex dim.")
    x = _lib.f32c(data)
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_complex_abs(_lib.ptr(x), _lib.ptr(out), x.numel() // 2, int(squared), _lib.stream_ptr()),
               "mrx_complex_abs")
    return out


def complex_abs(data: torch.Tensor) -> torch.Tensor:
    """utils.py:142-157."""
    return _abs(data, False)


def complex_abs_sq(data: torch.Tensor) -> torch.Tensor:
    """utils.py:160-175."""
    return _abs(data, True)


def check_stacked_complex(data: torch.Tensor) -> torch.Tensor:
    """utils.py:178-191."""
    return torch.view_as_complex(data) if data.shape[-1] == 2 else data


def rss(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """utils.py:194-209: sqrt((data**2).sum(dim)) on the tensor as given (real view: re/im are not combined)."""
    x = _lib.f32c(data)
    dim = dim % x.dim()
    outer, R, inner = _prod(x.shape[:dim]), int(x.shape[dim]), _prod(x.shape[dim + 1:])
    out = torch.empty(list(x.shape[:dim]) + list(x.shape[dim + 1:]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rss(_lib.ptr(x), _lib.ptr(out), outer, R, inner, _lib.stream_ptr()), "mrx_rss")
    return out


def rss_complex(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """utils.py:212-227."""
    if data.shape[-1] != 2:
        raise ValueError("Tensor does not have separate complex dim.")
    x = _lib.f32c(data)
    dim = dim % x.dim()
    if dim == x.dim() - 1:
        raise ValueError("rss_complex cannot reduce the complex dim")
    outer, R, inner = _prod(x.shape[:dim]), int(x.shape[dim]), _prod(x.shape[dim + 1:-1])
    out = torch.empty(list(x.shape[:dim]) + list(x.shape[dim + 1:-1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_rss_complex(_lib.ptr(x), _lib.ptr(out), outer, R, inner, _lib.stream_ptr()), "mrx_rss_complex")
    return out


def sense(data: torch.Tensor, sensitivity_maps: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """utils.py:230-248: complex_mul(data, conj(S)).sum(dim)."""
    if not data.shape[-1] == sensitivity_maps.shape[-1] == 2:
        raise ValueError("Tensors do not have separate complex dim.")
    x, s = _lib.f32c(data), _lib.f32c(sensitivity_maps)
    if x.shape != s.shape:
        shape = torch.broadcast_shapes(x.shape, s.shape)
        x, s = x.expand(shape).contiguous(), s.expand(shape).contiguous()
    dim = dim % x.dim()
    if dim == x.dim() - 1:
        raise ValueError("sense cannot reduce the complex dim")
    outer, R, inner = _prod(x.shape[:dim]), int(x.shape[dim]), _prod(x.shape[dim + 1:-1])
    out = torch.empty(list(x.shape[:dim]) + list(x.shape[dim + 1:]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mrx_sense(_lib.ptr(x), _lib.ptr(s), _lib.ptr(out), outer, R, inner, _lib.stream_ptr()), "mrx_sense")
    return out


def coil_combination(data: torch.Tensor, sensitivity_maps: torch.Tensor, method: str = "SENSE", dim: int = 0) -> torch.Tensor:
    """utils.py:251-272."""
    if method == "SENSE":
        return sense(data, sensitivity_maps, dim)
    if method == "RSS":
        return rss(data, dim)
    raise ValueError("Output type not supported.")


def save_reconstructions(reconstructions, out_dir):
    """utils.py:275-290: one `<out_dir>/<fname>` HDF5 file per volume with a `reconstruction` dataset (the leaderboard layout).
    Written by this package's HDF5 writer (`h5lite`); h5py, h5dump and MATLAB read the result."""
    from pathlib import Path

    from mridc_amd.collections.common.parts import h5lite
    out_dir = Path(out_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    for fname, recons in reconstructions.items():
        with h5lite.File(out_dir / fname, "w") as hf:
            hf.create_dataset("reconstruction", data=recons)


def apply_mask(data: torch.Tensor, mask_func=None, seed=None, padding: Optional[Sequence[int]] = None, shift: bool = False,
               half_scan_percentage: Optional[float] = 0.0, center_scale: Optional[float] = 0.02,
               existing_mask: Optional[torch.Tensor] = None) -> Tuple[Any, Any, int]:
    """utils.py:293-343.  The mask comes from the caller's `mask_func` (host object) or `existing_mask`;
    the arithmetic `data * mask + 0.0` runs on the GPU.  data: [..., H, W, 2]."""
    _lib.require_gpu(data)
    shape = np.array(data.shape)
    shape[:-3] = 1
    if existing_mask is None:
        mask, acc = mask_func(shape, seed, half_scan_percentage=half_scan_percentage, scale=center_scale)
    else:
        mask = existing_mask
        acc = mask.numel() / mask.sum()
    mask = mask.to(data.device).float().clone()
    if padding is not None and padding[0] != 0:
        mask[:, :, : padding[0]] = 0
        mask[:, :, padding[1]:] = 0
    if shift:
        from .fft import fftshift
        mask = fftshift(mask, dim=(1, 2))
    x = _lib.f32c(data)
    H, W = int(x.shape[-3]), int(x.shape[-2])
    lead = _prod(x.shape[:-3])
    m = mask.reshape(-1, mask.shape[-3], mask.shape[-2]) if mask.dim() >= 3 else mask.reshape(1, 1, -1)
    if m.shape[0] not in (1, lead) or m.shape[1] not in (1, H) or m.shape[2] not in (1, W):
        raise ValueError(f"mask of shape {tuple(mask.shape)} does not broadcast to data {tuple(data.shape)}")
    m = m.contiguous()
    st = [m.stride(0) if m.shape[0] != 1 else 0, 0, m.stride(1) if m.shape[1] != 1 else 0, m.stride(2) if m.shape[2] != 1 else 0]
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mrx_apply_mask(_lib.ptr(x), _lib.ptr(m), _lib.ptr(out), lead, 1, H, W, _lib.i64_array(st),
                                         _lib.stream_ptr()), "mrx_apply_mask")
    return out, mask, acc


def mask_center(x: torch.Tensor, mask_from: Optional[int], mask_to: Optional[int], mask_type: str = "2D") -> torch.Tensor:
    """utils.py:346-376 (index-only: slice copy)."""
    mask = torch.zeros_like(x)
    if isinstance(mask_from, list):
        mask_from = mask_from[0]
    if isinstance(mask_to, list):
        mask_to = mask_to[0]
    if mask_type == "1D":
        mask[:, :, :, mask_from:mask_to] = x[:, :, :, mask_from:mask_to]
    elif mask_type == "2D":
        mask[:, :, mask_from:mask_to] = x[:, :, mask_from:mask_to]
    return mask


def batched_mask_center(x: torch.Tensor, mask_from: torch.Tensor, mask_to: torch.Tensor, mask_type: str = "2D") -> torch.Tensor:
    """utils.py:379-410."""
    if mask_from.shape != mask_to.shape:
        raise ValueError("mask_from and mask_to must match shapes.")
    if mask_from.ndim != 1:
        raise ValueError("mask_from and mask_to must have 1 dimension.")
    if mask_from.shape[0] not in (1, x.shape[0]) or x.shape[0] != mask_to.shape[0]:
        raise ValueError("mask_from and mask_to must have batch_size length.")
    if mask_from.shape[0] == 1:
        return mask_center(x, int(mask_from), int(mask_to), mask_type=mask_type)
    mask = torch.zeros_like(x)
    for i, (start, end) in enumerate(zip(mask_from, mask_to)):
        mask[i, :, :, start:end] = x[i, :, :, start:end]
    return mask


def center_crop(data: torch.Tensor, shape: Tuple[int, int]) -> torch.Tensor:
    """utils.py:413-435 (a view)."""
    if not (0 < shape[0] <= data.shape[-2] and 0 < shape[1] <= data.shape[-1]):
        raise ValueError("Invalid shapes.")
    w_from = int((data.shape[-2] - shape[0]) / 2)
    h_from = int((data.shape[-1] - shape[1]) / 2)
    return data[..., w_from:w_from + shape[0], h_from:h_from + shape[1]]


def complex_center_crop(data: torch.Tensor, shape: Tuple[int, int]) -> torch.Tensor:
    """utils.py:438-460."""
    if not (0 < shape[0] <= data.shape[-3] and 0 < shape[1] <= data.shape[-2]):
        raise ValueError("Invalid shapes.")
    w_from = int((data.shape[-3] - shape[0]) / 2)
    h_from = int((data.shape[-2] - shape[1]) / 2)
    return data[..., w_from:w_from + shape[0], h_from:h_from + shape[1], :]


def center_crop_to_smallest(x, y):
    """utils.py:463-486."""
    smallest_width = min(x.shape[-1], y.shape[-1])
    smallest_height = min(x.shape[-2], y.shape[-2])
    return center_crop(x, (smallest_height, smallest_width)), center_crop(y, (smallest_height, smallest_width))
