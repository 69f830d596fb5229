"""Drop-in for `mridc.collections.reconstruction.models.cascadenet.ccnn_block.CascadeNetBlock` (reference ccnn_block.py:10-139)."""
from typing import Optional, Tuple

import torch

from mridc_amd import ops
from mridc_amd.collections.reconstruction.models.conv import conv2d


class CascadeNetBlock(torch.nn.Module):
    """Soft data consistency with the input model as regulariser: the same cascade as VarNetBlock (sens_reduce -> model ->
    sens_expand -> pred - soft_dc - eta, mrx_sens_reduce / mrx_sens_expand / mrx_dc_combine) around a channels-first model."""

    def __init__(self, model: torch.nn.Module, fft_centered: bool = True, fft_normalization: str = "ortho",
                 spatial_dims: Optional[Tuple[int, int]] = None, coil_dim: int = 1, no_dc: bool = False):
        super().__init__()
        self.model = model
        self.fft_centered = fft_centered
        self.fft_normalization = fft_normalization
        self.spatial_dims = spatial_dims if spatial_dims is not None else [-2, -1]
        self.coil_dim = coil_dim
        self.no_dc = no_dc
        self.dc_weight = torch.nn.Parameter(torch.ones(1))
        self._hybrid = False   # set by the model for row-invariant masks: k-space arguments are IFFT_H(k), row transforms only
        if coil_dim != 1:
            raise NotImplementedError("the HIP path expects the coil dimension at index 1")

    def sens_expand(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """ccnn_block.py:56-77."""
        return ops.sens_expand(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=self._hybrid)

    def sens_reduce(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """ccnn_block.py:79-99 (keepdim on the coil axis)."""
        return ops.sens_reduce(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims,
                               hybrid=self._hybrid).unsqueeze(1)

    def forward(self, pred: torch.Tensor, ref_kspace: torch.Tensor, sens_maps: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """ccnn_block.py:101-139."""
        # hybrid cascades chained by the model (W = 372): `_reduced_in` replaces this block's own sens_reduce, `_reduced_out` is the next one's
        eta = self._reduced_in.unsqueeze(1) if (self._hybrid and getattr(self, "_reduced_in", None) is not None) \
            else self.sens_reduce(pred, sens_maps)
        self._reduced_out = None
        chain = self._hybrid and getattr(self, "_want_reduced", False) and ops.sens_expand_dc_reduce_supported(sens_maps)
        x = eta.squeeze(self.coil_dim).permute(0, 3, 1, 2)
        if isinstance(self.model, conv2d.Conv2d):
            eta = self.model(x, _complex_last=True)            # the last conv writes [B,H,W,2] directly
        else:
            eta = self.model(x).permute(0, 2, 3, 1)
        if eta.dim() < sens_maps.dim():
            eta = eta.unsqueeze(1)
        if self._hybrid and not self.no_dc:      # expand + data consistency in one pass over the coil stack
            if chain:
                out, self._reduced_out = ops.sens_expand_dc_hybrid(eta, sens_maps, pred, ref_kspace, mask, self.dc_weight, self.fft_centered,
                                                                   self.fft_normalization, reduce=True)
                return out
            return ops.sens_expand_dc_hybrid(eta, sens_maps, pred, ref_kspace, mask, self.dc_weight, self.fft_centered,
                                             self.fft_normalization)
        if chain:                                # no_dc (the model-zoo configuration): expand, and the next block's reduction with it
            out, self._reduced_out = ops.sens_expand(eta, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=True,
                                                     reduce=True)
            return out
        eta = self.sens_expand(eta, sens_maps)
        if not self.no_dc:
            eta = ops.dc_combine(pred, pred, ref_kspace, mask, self.dc_weight, eta)   # pred - where(mask, pred - ref, 0) * w - eta
        return eta
