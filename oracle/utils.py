"""Oracle: complex / coil operators (reference mridc/collections/common/parts/utils.py).  Test infrastructure."""
import numpy as np
import torch


def _need_complex_dim(*ts):
    for t in ts:
        if t.shape[-1] != 2:
            raise ValueError("Tensor does not have separate complex dim.")


def to_tensor(data):
    """utils.py:53-71."""
    if np.iscomplexobj(data):
        data = np.stack((data.real, data.imag), axis=-1)
    return torch.from_numpy(data)


def tensor_to_complex_np(data):
    """utils.py:74-88."""
    data = data.numpy()
    return data[..., 0] + 1j * data[..., 1]


def complex_mul(x, y):
    """utils.py:96-118."""
    if not x.shape[-1] == y.shape[-1] == 2:
        raise ValueError("Tensors do not have separate complex dim.")
    xr, xi = x[..., 0], x[..., 1]
    yr, yi = y[..., 0], y[..., 1]
    return torch.stack((xr * yr - xi * yi, xr * yi + xi * yr), dim=-1)


def complex_conj(x):
    """utils.py:121-139."""
    _need_complex_dim(x)
    return torch.stack((x[..., 0], -x[..., 1]), dim=-1)


def complex_abs_sq(data):
    """utils.py:160-175."""
    _need_complex_dim(data)
    return (data * data).sum(dim=-1)


def complex_abs(data):
    """utils.py:142-157."""
    return complex_abs_sq(data).sqrt()


def check_stacked_complex(data):
    """utils.py:178-191."""
    return torch.view_as_complex(data) if data.shape[-1] == 2 else data


def rss(data, dim=0):
    """utils.py:194-209.  NB: squares the real-view tensor; re/im are NOT combined (appendix D.11)."""
    return torch.sqrt((data * data).sum(dim))


def rss_complex(data, dim=0):
    """utils.py:212-227."""
    return torch.sqrt(complex_abs_sq(data).sum(dim))


def sense(data, sensitivity_maps, dim=0):
    """utils.py:230-248."""
    return complex_mul(data, complex_conj(sensitivity_maps)).sum(dim)


def coil_combination(data, sensitivity_maps, method="SENSE", dim=0):
    """utils.py:251-272."""
    if method == "SENSE":
        return sense(data, sensitivity_maps, dim)
    if method == "RSS":
        return rss(data, dim)
    raise ValueError("Output type not supported.")


def center_crop(data, shape):
    """utils.py:413-435 (window start = (dim - size) // 2, truncating)."""
    if not (0 < shape[0] <= data.shape[-2] and 0 < shape[1] <= data.shape[-1]):
        raise ValueError("Invalid shapes.")
    r0 = int((data.shape[-2] - shape[0]) / 2)
    c0 = int((data.shape[-1] - shape[1]) / 2)
    return data[..., r0:r0 + shape[0], c0:c0 + shape[1]]


def complex_center_crop(data, shape):
    """utils.py:438-460."""
    if not (0 < shape[0] <= data.shape[-3] and 0 < shape[1] <= data.shape[-2]):
        raise ValueError("Invalid shapes.")
    r0 = int((data.shape[-3] - shape[0]) / 2)
    c0 = int((data.shape[-2] - shape[1]) / 2)
    return data[..., r0:r0 + shape[0], c0:c0 + shape[1], :]


def center_crop_to_smallest(x, y):
    """utils.py:463-486."""
    w = min(x.shape[-1], y.shape[-1])
    h = min(x.shape[-2], y.shape[-2])
    return center_crop(x, (h, w)), center_crop(y, (h, w))


def apply_existing_mask(data, mask, padding=None, shift=False):
    """utils.py:325-343 with `existing_mask` given (mask generation itself is a 'next' row, N2).

    Returns (masked_data, mask, acc).  `data*mask + 0.0` removes negative zeros (appendix D.12).
    """
    acc = mask.numel() / mask.sum()
    mask = mask.clone()
    if padding is not None and padding[0] != 0:
        mask[:, :, : padding[0]] = 0
        mask[:, :, padding[1]:] = 0
    if shift:
        mask = torch.fft.fftshift(mask, dim=(1, 2))
    return data * mask + 0.0, mask, acc
