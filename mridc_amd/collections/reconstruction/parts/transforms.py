"""Drop-in for `mridc.collections.reconstruction.parts.transforms.MRIDataTransforms` (reference parts/transforms.py:17-619) with
the per-sample preprocessing ON THE DEVICE.

The reference prepares every slice in CPU data workers: target = |SENSE(ifft2(kspace))| / max, (optional) centre crop through
image space, masking, and max-normalisation of `kspace` and `y` -- each an `ifft2 -> / max -> fft2` pair over the full coil stack --
plus the map / target scalings.  That is 4-6 full-coil transforms per slice; with the cascades at ~15 ms per slice it would bound an
8-GPU node on host cores.  Here the same steps run on the HIP operators (`mrx_fft2`, `mrx_sense`/`mrx_rss`, `mrx_apply_mask`) and two
small reductions (`mrx_max_abs`, `mrx_div_by_device_scalar`) whose scalar never leaves the device, so the chain is sync-free.

Same constructor, same call signature, same 9-tuple; inputs may be the NumPy complex arrays the reference's dataset yields or
real-view tensors that are already on the GPU.  Not carried over (raise `NotImplementedError` at construction): noise pre-whitening,
geometric-decomposition coil compression, k-space zero filling, 3-D data -- none of them is used by the configs of SURVEY section 6.
Masks come from `mridc_amd.collections.reconstruction.data.subsample` (bit-identical to the reference's) or from the caller.
"""
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
import mridc_amd.collections.reconstruction.data.subsample as subsample
from mridc_amd import ops

__all__ = ["MRIDataTransforms"]


def _unset(v) -> bool:
    return v is None or v in ("", "None")


class MRIDataTransforms:
    """MRI preprocessing data transforms, device-resident."""

    def __init__(self, apply_prewhitening: bool = False, prewhitening_scale_factor: float = 1.0, prewhitening_patch_start: int = 10,
                 prewhitening_patch_length: int = 30, apply_gcc: bool = False, gcc_virtual_coils: int = 10, gcc_calib_lines: int = 24,
                 gcc_align_data: bool = True, coil_combination_method: str = "SENSE", dimensionality: int = 2,
                 mask_func: Optional[List[subsample.MaskFunc]] = None, shift_mask: bool = False,
                 mask_center_scale: Optional[float] = 0.02, half_scan_percentage: float = 0.0, remask: bool = False,
                 crop_size: Optional[Tuple[int, int]] = None, kspace_crop: bool = False, crop_before_masking: bool = True,
                 kspace_zero_filling_size: Optional[Tuple] = None, normalize_inputs: bool = False, fft_centered: bool = True,
                 fft_normalization: str = "ortho", max_norm: bool = True, spatial_dims: Sequence[int] = None, coil_dim: int = 0,
                 use_seed: bool = True, device: Union[str, torch.device] = "cuda"):
        if apply_prewhitening or apply_gcc:
            raise NotImplementedError("mridc_amd.MRIDataTransforms: pre-whitening / coil compression are not part of the device path")
        if not _unset(kspace_zero_filling_size):
            raise NotImplementedError("mridc_amd.MRIDataTransforms: kspace_zero_filling_size is not part of the device path")
        if dimensionality != 2:
            raise NotImplementedError("mridc_amd.MRIDataTransforms implements dimensionality=2")
        self.coil_combination_method = coil_combination_method
        self.dimensionality = dimensionality
        self.mask_func = mask_func
        self.shift_mask = shift_mask
        self.mask_center_scale = mask_center_scale
        self.half_scan_percentage = half_scan_percentage
        self.remask = remask
        self.crop_size = crop_size
        self.kspace_crop = kspace_crop
        self.crop_before_masking = crop_before_masking
        self.kspace_zero_filling_size = kspace_zero_filling_size
        self.normalize_inputs = normalize_inputs
        self.fft_centered = fft_centered
        self.fft_normalization = fft_normalization
        self.max_norm = max_norm
        self.spatial_dims = spatial_dims if spatial_dims is not None else [-2, -1]
        self.coil_dim = coil_dim - 1                                  # transforms.py:137 (2-D: the batch dim is not there yet)
        self.apply_prewhitening = False
        self.prewhitening = None
        self.gcc = None
        self.use_seed = use_seed
        self.device = torch.device(device)

    # ---- helpers ----------------------------------------------------------------------------------------------------------
    def _dev(self, x) -> torch.Tensor:
        """NumPy complex array or real-view tensor -> fp32 real-view tensor on the device."""
        if isinstance(x, np.ndarray):
            x = utils.to_tensor(x)
        return x.to(self.device, dtype=torch.float32)

    def _fft2(self, x):
        return fft.fft2(x, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims)

    def _ifft2(self, x):
        return fft.ifft2(x, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims)

    def _max_normalise_kspace(self, k):
        """`ifft2 -> / max|.| -> fft2` (transforms.py:527-617); max over every real and imaginary component, as the reference's
        torch.max(torch.abs(real view)).  fft_normalization "none": unnormalised transforms, complex modulus."""
        if self.fft_normalization in ("backward", "ortho", "forward"):
            im = self._ifft2(k)
            if self.max_norm:
                im = ops.div_by_device_scalar(im, ops.max_abs(im))
            return self._fft2(im)
        if self.fft_normalization in ("none", None) and self.max_norm:
            im = fft.ifft2(k, centered=False, normalization="backward", spatial_dims=self.spatial_dims)   # == torch.fft.ifftn(norm=None)
            im = ops.div_by_device_scalar(im, ops.max_abs(im, complex_modulus=True))
            return fft.fft2(im, centered=False, normalization="backward", spatial_dims=self.spatial_dims)
        return k

    # ---- the transform --------------------------------------------------------------------------------------------------------
    def __call__(self, kspace, sensitivity_map, mask, eta, target, attrs: Dict, fname: str, slice_idx: int) -> Tuple[
            torch.Tensor, Union[List, torch.Tensor], Any, Union[List, Any], Any, torch.Tensor, str, int, Union[List, Any]]:
        """transforms.py:155-619.  Returns (kspace, masked_kspace, sensitivity_map, mask, eta, target, fname, slice_idx, acc)."""
        kspace = self._dev(kspace)
        have_maps = sensitivity_map is not None and (sensitivity_map.size != 0 if isinstance(sensitivity_map, np.ndarray)
                                                     else sensitivity_map.numel() != 0)
        if have_maps:
            sensitivity_map = self._dev(sensitivity_map)
        eta_given = eta is not None and (eta.size != 0 if isinstance(eta, np.ndarray) else eta.numel() != 0)
        eta = self._dev(eta) if eta_given else torch.tensor([])

        # target (transforms.py:258-288): coil-combined reference image, magnitude, scaled to max 1
        method = self.coil_combination_method.upper()
        if method == "RSS":
            target = utils.rss(self._ifft2(kspace), dim=self.coil_dim)
        elif method == "SENSE":
            if have_maps:
                target = utils.sense(self._ifft2(kspace), sensitivity_map, dim=self.coil_dim)
        elif target is not None and np.size(target) != 0:
            target = self._dev(target)
        elif "target" in attrs or "target_rss" in attrs:
            target = torch.tensor(attrs["target"]).to(self.device)
        else:
            raise ValueError("No target found")
        target = ops.div_by_device_scalar(target, ops.max_abs(target, complex_modulus=True), modulus=True)

        seed = tuple(map(ord, fname)) if self.use_seed else None
        acq_start = attrs["padding_left"] if "padding_left" in attrs else 0
        acq_end = attrs["padding_right"] if "padding_left" in attrs else 0

        cropping = not _unset(self.crop_size)
        if cropping:                                                  # transforms.py:296-349
            h = min(int(self.crop_size[0]), target.shape[0])
            w = min(int(self.crop_size[1]), target.shape[1])
            self.crop_size = (int(h), int(w))                          # (the reference keeps the clipped size on the object too)
            target = utils.center_crop(target, self.crop_size)
            if have_maps:
                sensitivity_map = (self._ifft2(utils.complex_center_crop(self._fft2(sensitivity_map), self.crop_size))
                                   if self.kspace_crop else utils.complex_center_crop(sensitivity_map, self.crop_size))
            if eta_given and eta.ndim > 2:
                eta = (self._ifft2(utils.complex_center_crop(self._fft2(eta), self.crop_size))
                       if self.kspace_crop else utils.complex_center_crop(eta, self.crop_size))

        def crop_kspace(k):                                            # image-space crop unless kspace_crop (transforms.py:352-370)
            return (utils.complex_center_crop(k, self.crop_size) if self.kspace_crop
                    else self._fft2(utils.complex_center_crop(self._ifft2(k), self.crop_size)))

        if cropping and self.crop_before_masking:
            kspace = crop_kspace(kspace)

        if not utils.is_none(mask):                                    # a stored mask (transforms.py:372-396)
            for _mask in mask:
                if list(_mask.shape) == [kspace.shape[-3], kspace.shape[-2]]:
                    mask = torch.from_numpy(np.asarray(_mask)).unsqueeze(0).unsqueeze(-1)
                    break
            if isinstance(mask, np.ndarray):
                mask = torch.from_numpy(mask).unsqueeze(0).unsqueeze(-1)
            mask = mask.to(self.device)
            if not utils.is_none(acq_start) and not utils.is_none(acq_end) and acq_start != 0:
                mask = mask.clone()
                mask[:, :, :acq_start] = 0
                mask[:, :, acq_end:] = 0
            if self.shift_mask:
                mask = fft.fftshift(mask, dim=(self.spatial_dims[0] - 1, self.spatial_dims[1] - 1))
            if cropping and self.crop_before_masking:
                mask = utils.complex_center_crop(mask, self.crop_size)
            masked_kspace, _, _ = utils.apply_mask(kspace, existing_mask=mask)
            acc = 1
        elif utils.is_none(self.mask_func):                            # fully sampled (transforms.py:397-424 with mask None)
            masked_kspace = kspace.clone()
            acc = torch.tensor([1])
            mask = torch.ones(masked_kspace.shape[-3], masked_kspace.shape[-2], dtype=torch.float32, device=self.device)
            if cropping:
                mask = utils.center_crop(mask, self.crop_size)
            mask = mask.unsqueeze(0).unsqueeze(-1)
            if self.shift_mask:
                mask = fft.fftshift(mask, dim=(1, 2))
            masked_kspace, _, _ = utils.apply_mask(masked_kspace, existing_mask=mask)
            mask = mask.byte()
        elif isinstance(self.mask_func, list):                         # one masked copy per mask function (transforms.py:425-467)
            masked_kspace, mask, acc = [], [], []
            for m in self.mask_func:
                _y, _m, _a = utils.apply_mask(kspace, m, seed, (acq_start, acq_end), shift=self.shift_mask,
                                              half_scan_percentage=self.half_scan_percentage, center_scale=self.mask_center_scale)
                masked_kspace.append(_y)
                mask.append(_m.byte())
                acc.append(_a)
        else:
            masked_kspace, mask, acc = utils.apply_mask(kspace, self.mask_func[0], seed, (acq_start, acq_end), shift=self.shift_mask,
                                                        half_scan_percentage=self.half_scan_percentage,
                                                        center_scale=self.mask_center_scale)
            mask = mask.byte()

        if cropping and not self.crop_before_masking:                  # transforms.py:480-524
            kspace = crop_kspace(kspace)
            if isinstance(masked_kspace, list):
                masked_kspace = [crop_kspace(y) for y in masked_kspace]
                mask = [utils.center_crop(m.squeeze(-1), self.crop_size).unsqueeze(-1) for m in mask]
            else:
                masked_kspace = crop_kspace(masked_kspace)
                mask = utils.center_crop(mask.squeeze(-1), self.crop_size).unsqueeze(-1)

        if self.normalize_inputs:                                      # transforms.py:527-617
            kspace = self._max_normalise_kspace(kspace)
            if isinstance(masked_kspace, list):
                masked_kspace = [self._max_normalise_kspace(y) for y in masked_kspace]
            else:
                masked_kspace = self._max_normalise_kspace(masked_kspace)
            if self.max_norm:
                if have_maps:
                    sensitivity_map = ops.div_by_device_scalar(sensitivity_map, ops.max_abs(sensitivity_map))
                if eta_given and eta.ndim > 2:
                    eta = ops.div_by_device_scalar(eta, ops.max_abs(eta))
                target = ops.div_by_device_scalar(target, ops.max_abs(target))
        return kspace, masked_kspace, sensitivity_map, mask, eta, target, fname, slice_idx, acc
