"""Drop-in for `mridc.collections.reconstruction.models.varnet.vn_block.VarNetBlock` (reference vn_block.py:12-119)."""
from typing import Optional, Tuple

import torch

from mridc_amd import diff, ops


class VarNetBlock(torch.nn.Module):
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
        # hybrid cascades chained by the model (W = 372): the data-consistency pass of one block also produces the sens_reduce the next one
        # starts with (mrx_pfa372_expand_reduce): `_reduced_in` replaces this block's own sens_reduce, `_reduced_out` is what it hands on
        self._reduced_in = None
        self._want_reduced = False
        self._reduced_out = None
        if coil_dim != 1:
            raise NotImplementedError("the HIP path expects the coil dimension at index 1")

    def sens_expand(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """vn_block.py:51-69 (mrx_sens_expand)."""
        return ops.sens_expand(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims, hybrid=self._hybrid)

    def sens_reduce(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        """vn_block.py:71-87 (mrx_sens_reduce), keepdim on the coil axis."""
        return ops.sens_reduce(x, sens_maps, self.fft_centered, self.fft_normalization, self.spatial_dims,
                               hybrid=self._hybrid).unsqueeze(1)

    def forward(self, pred: torch.Tensor, ref_kspace: torch.Tensor, sens_maps: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """vn_block.py:89-119."""
        if diff.active(pred, *self.parameters(), training=self.training):            # training: the differentiable forms (k-space formulation, no fusion)
            kw = (self.fft_centered, self.fft_normalization, self.spatial_dims)
            eta = diff.sens_reduce(pred, sens_maps, *kw).unsqueeze(1)
            eta = diff.sens_expand(self.model(eta), sens_maps, *kw)
            return eta if self.no_dc else diff.dc_combine(pred, pred, ref_kspace, mask, self.dc_weight, eta)
        eta = self._reduced_in.unsqueeze(1) if (self._hybrid and self._reduced_in is not None) else self.sens_reduce(pred, sens_maps)
        self._reduced_out = None
        eta = self.model(eta)
        if self._hybrid and not self.no_dc:      # expand + data consistency in one pass over the coil stack
            if self._want_reduced and ops.sens_expand_dc_reduce_supported(sens_maps):
                out, self._reduced_out = ops.sens_expand_dc_hybrid(eta, sens_maps, pred, ref_kspace, mask, self.dc_weight, self.fft_centered,
                                                                   self.fft_normalization, reduce=True)
                return out
            return ops.sens_expand_dc_hybrid(eta, sens_maps, pred, ref_kspace, mask, self.dc_weight, self.fft_centered,
                                             self.fft_normalization)
        eta = self.sens_expand(eta, sens_maps)
        if not self.no_dc:
            # pred - where(mask.bool(), pred - ref, 0) * dc_weight - eta, one launch
            eta = ops.dc_combine(pred, pred, ref_kspace, mask, self.dc_weight, eta)
        return eta
