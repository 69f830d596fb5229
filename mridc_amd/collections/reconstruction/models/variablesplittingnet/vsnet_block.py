"""Drop-in for `mridc.collections.reconstruction.models.variablesplittingnet.vsnet_block` (reference vsnet_block.py:12-146)."""
from typing import Any, List, Optional, Tuple, Union

import torch

from mridc_amd import ops
from mridc_amd.collections.reconstruction.models.conv import conv2d


class DataConsistencyLayer(torch.nn.Module):
    """Hard replacement of the sampled locations (vsnet_block.py:12-25): ((1 - mask) * pred + mask * ref) * dc_weight."""

    def __init__(self):
        super().__init__()
        self.dc_weight = torch.nn.Parameter(torch.ones(1))

    def forward(self, pred_kspace, ref_kspace, mask):
        return ops.hard_dc(pred_kspace, ref_kspace, mask, self.dc_weight)


class WeightedAverageTerm(torch.nn.Module):
    """vsnet_block.py:28-36: param * x + (1 - param) * Sx."""

    def __init__(self):
        super().__init__()
        self.param = torch.nn.Parameter(torch.ones(1))

    def forward(self, x, Sx):
        if x.dim() == 5 and Sx.dim() == 4 and x.shape[0] == 1:
            return ops.vs_average(x, torch.zeros_like(x), Sx, self.param)
        if x.shape == Sx.shape and x.dim() == 5:
            # same-shape form: fold Sx's coil axis into the batch axis (one image per (b, c))
            B, C, H, W, _ = x.shape
            return ops.vs_average(x.reshape(B * C, 1, H, W, 2), torch.zeros_like(x).reshape(B * C, 1, H, W, 2),
                                  Sx.reshape(B * C, H, W, 2), self.param).reshape(x.shape)
        raise NotImplementedError(f"WeightedAverageTerm: shapes {tuple(x.shape)} / {tuple(Sx.shape)} are not on the HIP path")


class VSNetBlock(torch.nn.Module):
    """vsnet_block.py:39-146.  Per cascade: sens_reduce -> denoiser -> sens_expand -> hard DC -> sens_reduce -> weighted average
    (mrx_sens_reduce, conv kernels, mrx_sens_expand, mrx_hard_dc, mrx_sens_reduce, mrx_vs_average)."""

    def __init__(self, denoiser_block: torch.nn.ModuleList, data_consistency_block: torch.nn.ModuleList,
                 weighted_average_block: torch.nn.ModuleList, num_cascades: int = 8, fft_centered: bool = True,
                 fft_normalization: str = "ortho", spatial_dims: Optional[Tuple[int, int]] = None, coil_dim: int = 1):
        super().__init__()
        self.denoiser_block = denoiser_block
        self.data_consistency_block = data_consistency_block
        self.weighted_average_block = weighted_average_block
        self.num_cascades = num_cascades
        self.fft_centered = fft_centered
        self.fft_normalization = fft_normalization
        self.spatial_dims = spatial_dims if spatial_dims is not None else [-2, -1]
        self.coil_dim = coil_dim
        self._hybrid = False   # set by VSNet for row-invariant masks: k-space arguments are IFFT_H(k), row transforms only
        if coil_dim != 1:
            raise NotImplementedError("the HIP path expects the coil dimension at index 1")

    def sens_expand(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """vsnet_block.py:82-101.  The reference broadcasts a [B,H,W,2] image against [B,C,H,W,2] maps, which is only defined
        for batch 1 (the leading image axis lines up with the coil axis)."""
        if x.dim() == 4 and x.shape[0] != 1:
            raise NotImplementedError("VSNetBlock: the reference's [B,H,W,2] x [B,C,H,W,2] broadcast is only defined for batch 1 "
                                      "(or batch == coils, which is not reproduced)")
        return ops.sens_expand(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=self._hybrid)

    def sens_reduce(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """vsnet_block.py:103-116 (no keepdim)."""
        return ops.sens_reduce(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=self._hybrid)

    def forward(self, kspace: torch.Tensor, sens_maps: torch.Tensor, mask: torch.Tensor) -> List[Union[torch.Tensor, Any]]:
        """vsnet_block.py:118-146."""
        for idx in range(self.num_cascades):
            pred = self.sens_reduce(kspace, sens_maps)
            den = self.denoiser_block[idx]
            if isinstance(den, conv2d.Conv2d):
                pred = den(pred.permute(0, 3, 1, 2), _complex_last=True)
            else:
                pred = den(pred.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            pred = self.sens_expand(pred, sens_maps)
            sx = self.data_consistency_block[idx](pred, kspace, mask)
            sx = self.sens_reduce(sx, sens_maps)
            wa = self.weighted_average_block[idx]
            if isinstance(wa, WeightedAverageTerm):
                kspace = ops.vs_average(kspace, pred, sx, wa.param)      # param * (kspace + pred) + (1 - param) * sx, one launch
            else:
                kspace = wa(kspace + pred, sx)
        return kspace
