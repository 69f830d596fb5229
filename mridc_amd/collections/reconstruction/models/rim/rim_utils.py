"""Drop-in for `mridc.collections.reconstruction.models.rim.rim_utils` (reference rim_utils.py:11-67)."""
from typing import Sequence

import torch

from mridc_amd import ops


def log_likelihood_gradient(eta: torch.Tensor, masked_kspace: torch.Tensor, sense: torch.Tensor, mask: torch.Tensor,
                            sigma: float, fft_centered: bool, fft_normalization: str, spatial_dims: Sequence[int],
                            coil_dim: int) -> torch.Tensor:
    """Three fused HIP launches (csrc/fft.hip: mrx_llg).  eta [B,H,W,2]; y,S [B,C,H,W,2] -> [B,4,H,W]."""
    if coil_dim == 0:          # rim_utils.py:41-42
        coil_dim += 1
    if coil_dim != 1:
        raise NotImplementedError("log_likelihood_gradient: the HIP path expects the coil dimension at index 1")
    return ops.llg(eta, masked_kspace, sense, mask, sigma, fft_centered, fft_normalization, spatial_dims)
