"""Oracle: per-sample preprocessing (reference reconstruction/parts/transforms.py:155-619, 2-D path without pre-whitening, coil
compression and zero filling).  TEST INFRASTRUCTURE ONLY -- functional torch-CPU restatement, pinned by tests/golden/g12_transforms.npz
(produced by the reference's own MRIDataTransforms) and used as the checker of mridc_amd's device-side transforms.
"""
import numpy as np
import torch

from . import fft as offt
from . import utils as outils


def _unset(v):
    return v is None or v in ("", "None")


def preprocess(kspace, sensitivity_map, mask=None, eta=None, *, fname="f", attrs=None, mask_func=None, coil_combination_method="SENSE",
               shift_mask=False, mask_center_scale=0.02, half_scan_percentage=0.0, crop_size=None, kspace_crop=False,
               crop_before_masking=True, normalize_inputs=False, fft_centered=True, fft_normalization="ortho", max_norm=True,
               spatial_dims=None, coil_dim=0, use_seed=True):
    """Returns (kspace, masked_kspace, sensitivity_map, mask, eta, target, acc) as the reference's __call__ does (minus the
    pass-through fname / slice_idx).  `coil_dim` is the constructor argument (the transform uses coil_dim - 1, transforms.py:137)."""
    attrs = attrs or {}
    spatial_dims = [-2, -1] if spatial_dims is None else list(spatial_dims)
    cd = coil_dim - 1
    f2 = lambda x: offt.fft2(x, fft_centered, fft_normalization, spatial_dims)      # noqa: E731
    i2 = lambda x: offt.ifft2(x, fft_centered, fft_normalization, spatial_dims)     # noqa: E731
    kspace = outils.to_tensor(kspace) if isinstance(kspace, np.ndarray) else kspace
    S = sensitivity_map
    if isinstance(S, np.ndarray):
        S = outils.to_tensor(S) if S.size else None
    eta = outils.to_tensor(eta) if isinstance(eta, np.ndarray) and eta.size else (eta if torch.is_tensor(eta) and eta.numel() else None)

    if coil_combination_method.upper() == "RSS":                       # transforms.py:258-288
        target = outils.rss(i2(kspace), dim=cd)
    else:
        target = outils.sense(i2(kspace), S, dim=cd)
    target = torch.view_as_complex(target.contiguous())
    target = torch.abs(target / torch.max(torch.abs(target)))

    seed = tuple(map(ord, fname)) if use_seed else None
    acq_start = attrs.get("padding_left", 0)
    acq_end = attrs["padding_right"] if "padding_left" in attrs else 0
    cropping = not _unset(crop_size)
    if cropping:                                                       # transforms.py:296-349
        crop_size = (min(int(crop_size[0]), target.shape[0]), min(int(crop_size[1]), target.shape[1]))
        target = outils.center_crop(target, crop_size)
        if S is not None:
            S = i2(outils.complex_center_crop(f2(S), crop_size)) if kspace_crop else outils.complex_center_crop(S, crop_size)
        if eta is not None and eta.ndim > 2:
            eta = i2(outils.complex_center_crop(f2(eta), crop_size)) if kspace_crop else outils.complex_center_crop(eta, crop_size)

    def crop_k(k):                                                     # transforms.py:352-370
        return outils.complex_center_crop(k, crop_size) if kspace_crop else f2(outils.complex_center_crop(i2(k), crop_size))

    if cropping and crop_before_masking:
        kspace = crop_k(kspace)

    def masked(k, m):
        return k * m + 0.0                                             # transforms.py:394 / utils.py:341

    if mask is not None and np.size(mask) != 0:                        # transforms.py:372-396
        for _m in mask:
            if list(_m.shape) == [kspace.shape[-3], kspace.shape[-2]]:
                mask = torch.from_numpy(np.asarray(_m)).unsqueeze(0).unsqueeze(-1)
                break
        if isinstance(mask, np.ndarray):
            mask = torch.from_numpy(mask).unsqueeze(0).unsqueeze(-1)
        if acq_start:
            mask = mask.clone()
            mask[:, :, :acq_start] = 0
            mask[:, :, acq_end:] = 0
        if shift_mask:
            mask = torch.fft.fftshift(mask, dim=(spatial_dims[0] - 1, spatial_dims[1] - 1))
        if cropping and crop_before_masking:
            mask = outils.complex_center_crop(mask, crop_size)
        y, acc = masked(kspace, mask), 1
    elif mask_func is None:                                            # fully sampled
        mask = torch.ones(kspace.shape[-3], kspace.shape[-2], dtype=torch.float32)
        if cropping:
            mask = outils.center_crop(mask, crop_size)
        mask = mask.unsqueeze(0).unsqueeze(-1)
        if shift_mask:
            mask = torch.fft.fftshift(mask, dim=(1, 2))
        y, acc = kspace.clone() * mask, torch.tensor([1])
        mask = mask.byte()
    else:                                                              # generated mask(s): utils.py:293-343
        funcs = mask_func if isinstance(mask_func, list) else [mask_func[0] if isinstance(mask_func, tuple) else mask_func]
        ys, ms, accs = [], [], []
        for fn in funcs:
            shape = np.array(kspace.shape)
            shape[:-3] = 1
            m, a = fn(shape, seed, half_scan_percentage=half_scan_percentage, scale=mask_center_scale)
            if acq_start:
                m[:, :, :acq_start] = 0
                m[:, :, acq_end:] = 0
            if shift_mask:
                m = torch.fft.fftshift(m, dim=(1, 2))
            ys.append(masked(kspace, m))
            ms.append(m.byte())
            accs.append(a)
        # a list of mask functions yields lists, even of one (transforms.py:425-467); any other container takes the first
        y, mask, acc = (ys, ms, accs) if isinstance(mask_func, list) else (ys[0], ms[0], accs[0])

    if cropping and not crop_before_masking:                           # transforms.py:480-524
        kspace = crop_k(kspace)
        if isinstance(y, list):
            y = [crop_k(v) for v in y]
            mask = [outils.center_crop(m.squeeze(-1), crop_size).unsqueeze(-1) for m in mask]
        else:
            y = crop_k(y)
            mask = outils.center_crop(mask.squeeze(-1), crop_size).unsqueeze(-1)

    def maxnorm(k):                                                    # transforms.py:527-617
        if fft_normalization in ("backward", "ortho", "forward"):
            im = i2(k)
            if max_norm:
                im = im / torch.max(torch.abs(im))
            return f2(im)
        if max_norm:
            im = torch.fft.ifftn(torch.view_as_complex(k.contiguous()), dim=list(spatial_dims), norm=None)
            im = im / torch.max(torch.abs(im))
            return torch.view_as_real(torch.fft.fftn(im, dim=list(spatial_dims), norm=None))
        return k

    if normalize_inputs:
        kspace = maxnorm(kspace)
        y = [maxnorm(v) for v in y] if isinstance(y, list) else maxnorm(y)
        if max_norm:
            if S is not None:
                S = S / torch.max(torch.abs(S))
            if eta is not None and eta.ndim > 2:
                eta = eta / torch.max(torch.abs(eta))
            target = target / torch.max(torch.abs(target))
    return kspace, y, S, mask, eta, target, acc
