"""Drop-in for `mridc.collections.reconstruction.models.zf.ZF` (reference zf.py:24-100), inference path."""
import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd.collections.reconstruction.models import _cfg

__all__ = ["ZF"]

from mridc_amd.collections.reconstruction.models.base import build_sens_net


class ZF(torch.nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        self.fft_centered = cfg_dict.get("fft_centered")
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.use_sens_net = cfg_dict.get("use_sens_net")
        if self.use_sens_net:                                          # models/base.py:81-95
            self.sens_net = build_sens_net(cfg_dict, self.fft_centered, self.fft_normalization, self.spatial_dims, self.coil_dim)

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, target: torch.Tensor = None):
        """zf.py:61-100."""
        pred = utils.coil_combination(
            fft.ifft2(y, centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims),
            sensitivity_maps, method=self.coil_combination_method.upper(), dim=self.coil_dim)
        pred = utils.check_stacked_complex(pred)
        _, pred = utils.center_crop_to_smallest(target, pred)
        return pred

    forward_step = forward
