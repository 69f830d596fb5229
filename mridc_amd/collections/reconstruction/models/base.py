"""`BaseSensitivityModel` of the reference's `mridc.collections.reconstruction.models.base` (base.py:715-932): learned estimation
of the coil sensitivity maps from the fully sampled centre of k-space.

centre lines of the masked k-space -> IFFT2 -> NormUnet on every coil image (coils moved to the batch dim) -> division by the
root-sum-of-squares over coils.  Every arithmetic step runs on the HIP operators (`mrx_fft2`, the NormUnet kernels,
`mrx_div_rss_complex`); locating the centre block is index arithmetic on the mask (torch, as in the reference).
The reference's `BaseMRIReconstructionModel` builds this module when `use_sens_net` is set and applies it in its
train / validation / test steps (base.py:81-95,234,300,392); `build_sens_net(cfg_dict)` below reads the same config keys, and the
model classes of this package attach it as `self.sens_net` with the reference's state_dict layout (`sens_net.norm_unet.unet.*`).
"""
from typing import Optional, Sequence, Tuple

import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd import _lib
from mridc_amd.collections.reconstruction.models.unet_base import unet_block

__all__ = ["BaseSensitivityModel", "build_sens_net"]


class BaseSensitivityModel(torch.nn.Module):
    """Model for learning sensitivity estimation from k-space data (base.py:715-932)."""

    def __init__(self, chans: int = 8, num_pools: int = 4, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0,
                 padding_size: int = 15, mask_type: str = "2D", fft_centered: bool = True, fft_normalization: str = "ortho",
                 spatial_dims: Sequence[int] = None, coil_dim: int = 1, normalize: bool = True, mask_center: bool = True):
        super().__init__()
        self.mask_type = mask_type
        self.norm_unet = unet_block.NormUnet(chans, num_pools, in_chans=in_chans, out_chans=out_chans, drop_prob=drop_prob,
                                             padding_size=padding_size, normalize=normalize)
        self.mask_center = mask_center
        self.fft_centered = fft_centered
        self.fft_normalization = fft_normalization
        self.spatial_dims = spatial_dims if spatial_dims is not None else [-2, -1]
        self.coil_dim = coil_dim
        self.normalize = normalize

    @staticmethod
    def chans_to_batch_dim(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
        b, c, h, w, comp = x.shape
        return x.view(b * c, 1, h, w, comp), b

    @staticmethod
    def batch_chans_to_chan_dim(x: torch.Tensor, batch_size: int) -> torch.Tensor:
        bc, _, h, w, comp = x.shape
        return x.view(batch_size, bc // batch_size, h, w, comp)

    @staticmethod
    def divide_root_sum_of_squares(x: torch.Tensor, coil_dim: int) -> torch.Tensor:
        """x / rss_complex(x, coil_dim) (base.py:824-840), one kernel."""
        if x.shape[-1] != 2:
            raise ValueError("Tensor does not have separate complex dim.")
        x = _lib.f32c(x)
        dim = coil_dim % x.dim()
        outer, R, inner = utils._prod(x.shape[:dim]), int(x.shape[dim]), utils._prod(x.shape[dim + 1:-1])
        out = torch.empty_like(x)
        _lib.check(_lib.lib().mrx_div_rss_complex(_lib.ptr(x), _lib.ptr(out), outer, R, inner, _lib.stream_ptr()), "mrx_div_rss_complex")
        return out

    @staticmethod
    def get_pad_and_num_low_freqs(mask: torch.Tensor, num_low_frequencies: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """base.py:842-884: length of the contiguous run of sampled columns around the centre (symmetric, at least 1)."""
        if num_low_frequencies is None or num_low_frequencies == 0:
            squeezed_mask = mask[:, 0, 0, :, 0].to(torch.int8)
            cent = squeezed_mask.shape[1] // 2
            left = torch.argmin(squeezed_mask[:, :cent].flip(1), dim=1)
            right = torch.argmin(squeezed_mask[:, cent:], dim=1)
            num_low_frequencies_tensor = torch.max(2 * torch.min(left, right), torch.ones_like(left))
        else:
            num_low_frequencies_tensor = num_low_frequencies * torch.ones(mask.shape[0], dtype=mask.dtype, device=mask.device)
        pad = torch.div(mask.shape[-2] - num_low_frequencies_tensor + 1, 2, rounding_mode="trunc")
        return pad, num_low_frequencies_tensor

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, num_low_frequencies: Optional[int] = None) -> torch.Tensor:
        """base.py:886-932.  masked_kspace [B,C,H,W,2], mask [B,1,H|1,W,1] -> maps [B,C,H,W,2]."""
        _lib.require_gpu(masked_kspace)
        if self.mask_center:
            pad, num_low_freqs = self.get_pad_and_num_low_freqs(mask, num_low_frequencies)
            masked_kspace = utils.batched_mask_center(masked_kspace, pad, pad + num_low_freqs, mask_type=self.mask_type)
        images, batches = self.chans_to_batch_dim(
            fft.ifft2(masked_kspace, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims))
        images = self.batch_chans_to_chan_dim(self.norm_unet(images), batches)
        if self.normalize:
            images = self.divide_root_sum_of_squares(images, self.coil_dim)
        return images


def build_sens_net(cfg_dict, fft_centered, fft_normalization, spatial_dims, coil_dim):
    """The construction at base.py:81-95 (None when `use_sens_net` is off)."""
    if not cfg_dict.get("use_sens_net"):
        return None
    return BaseSensitivityModel(cfg_dict.get("sens_chans"), cfg_dict.get("sens_pools"), fft_centered=fft_centered,
                                fft_normalization=fft_normalization, spatial_dims=spatial_dims, coil_dim=coil_dim,
                                mask_type=cfg_dict.get("sens_mask_type"), normalize=cfg_dict.get("sens_normalize"),
                                mask_center=cfg_dict.get("sens_mask_center"))
