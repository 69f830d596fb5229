"""Drop-in for `mridc.collections.reconstruction.models.rvn.RecurrentVarNet` (reference rvn.py:24-226), inference path."""
import math
from typing import Optional

import torch

import mridc_amd.collections.common.parts.fft as fft
import mridc_amd.collections.common.parts.utils as utils
from mridc_amd import ops
from mridc_amd.collections.reconstruction.models import _cfg
from mridc_amd.collections.reconstruction.models.recurrentvarnet import recurrentvarnet

__all__ = ["RecurrentVarNet"]


class RecurrentVarNet(torch.nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg_dict = _cfg.to_dict(cfg)
        self.in_channels = cfg_dict.get("in_channels")
        self.recurrent_hidden_channels = cfg_dict.get("recurrent_hidden_channels")
        self.recurrent_num_layers = cfg_dict.get("recurrent_num_layers")
        self.no_parameter_sharing = cfg_dict.get("no_parameter_sharing")
        self.num_steps = 8 * math.ceil(cfg_dict.get("num_steps") / 8)      # rvn.py:51
        self.learned_initializer = cfg_dict.get("learned_initializer")
        self.initializer_initialization = cfg_dict.get("initializer_initialization")
        self.initializer_channels = cfg_dict.get("initializer_channels")
        self.initializer_dilations = cfg_dict.get("initializer_dilations")
        if (self.learned_initializer and self.initializer_initialization is not None and self.initializer_channels is not None
                and self.initializer_dilations is not None):
            if self.initializer_initialization not in ["sense", "input_image", "zero_filled"]:
                raise ValueError("Unknown initializer_initialization. Expected `sense`, `'input_image` or `zero_filled`."
                                 f"Got {self.initializer_initialization}.")
            self.initializer = recurrentvarnet.RecurrentInit(self.in_channels, self.recurrent_hidden_channels,
                                                             channels=self.initializer_channels, dilations=self.initializer_dilations,
                                                             depth=self.recurrent_num_layers,
                                                             multiscale_depth=cfg_dict.get("initializer_multiscale"))
        else:
            self.initializer = None
        self.fft_centered = cfg_dict.get("fft_centered")
        self.fft_normalization = cfg_dict.get("fft_normalization")
        self.spatial_dims = cfg_dict.get("spatial_dims")
        self.coil_dim = cfg_dict.get("coil_dim")
        self.coil_combination_method = cfg_dict.get("coil_combination_method")
        self.block_list = torch.nn.ModuleList()
        for _ in range(self.num_steps if self.no_parameter_sharing else 1):
            self.block_list.append(recurrentvarnet.RecurrentVarNetBlock(
                in_channels=self.in_channels, hidden_channels=self.recurrent_hidden_channels, num_layers=self.recurrent_num_layers,
                fft_centered=self.fft_centered, fft_normalization=self.fft_normalization, spatial_dims=self.spatial_dims,
                coil_dim=self.coil_dim))
        # rvn.py:104-108 applies rnn_weights_init, which only touches Linear / Embedding / LayerNorm modules (none here)
        self.train_loss_fn = _cfg.make_loss(cfg_dict.get("train_loss_fn", "l1"))
        self.val_loss_fn = _cfg.make_loss(cfg_dict.get("val_loss_fn", "l1"))
        self.accumulate_estimates = False

    def forward(self, y: torch.Tensor, sensitivity_maps: torch.Tensor, mask: torch.Tensor, init_pred: torch.Tensor,
                target: torch.Tensor, **kwargs) -> torch.Tensor:
        """rvn.py:128-226."""
        kw = dict(centered=self.fft_centered, normalization=self.fft_normalization, spatial_dims=self.spatial_dims)
        previous_state: Optional[list] = None
        if self.initializer is not None:
            if self.initializer_initialization == "sense":
                image = ops.sens_reduce(y, sensitivity_maps, self.fft_centered, self.fft_normalization, self.spatial_dims)
                initializer_input_image = image.unsqueeze(self.coil_dim)
            elif self.initializer_initialization == "input_image":
                if "initial_image" not in kwargs:
                    raise ValueError("`'initial_image` is required as input if initializer_initialization "
                                     f"is {self.initializer_initialization}.")
                initializer_input_image = kwargs["initial_image"].unsqueeze(self.coil_dim)
            else:
                initializer_input_image = fft.ifft2(y, **kw)
            k = fft.fft2(initializer_input_image, **kw)
            k = k[:, 0] if k.shape[1] == 1 else ops.coil_sum(k)           # .sum(1)
            previous_state = self.initializer(k.permute(0, 3, 1, 2), as_list=True)
        if previous_state is None:
            previous_state = [None] * self.recurrent_num_layers          # zero state, kept as a per-layer list between the blocks
        import os
        hybrid = (os.environ.get("MRIDC_AMD_HYBRID", "1") != "0" and self.coil_dim == 1
                  and str(self.coil_combination_method).upper() == "SENSE" and ops.mask_is_row_invariant(mask))
        # row-invariant mask: the k-space update commutes with the H transform -- the steps run on IFFT_H(k) with row transforms only
        y_k = ops.llg_prepare(y, self.fft_centered, self.fft_normalization, self.spatial_dims) if hybrid else y
        kspace_prediction = y_k.clone()
        red = None                               # sum_c conj(S) IFFT_W(k), handed from one step's update pass to the next step (W = 372)
        chain = os.environ.get("MRIDC_AMD_CHAIN_REDUCE", "1") != "0"
        for step in range(self.num_steps):
            block = self.block_list[step] if self.no_parameter_sharing else self.block_list[0]
            block._hybrid, block._reduced_in, block._want_reduced = hybrid, red, hybrid and chain
            try:
                kspace_prediction, previous_state = block(kspace_prediction, y_k, mask, sensitivity_maps, previous_state)
                red = block._reduced_out
            finally:
                block._hybrid, block._reduced_in, block._want_reduced, block._reduced_out = False, None, False, None
        if hybrid and red is not None:
            eta = red
        elif hybrid:
            eta = ops.sens_reduce(kspace_prediction, sensitivity_maps, self.fft_centered, self.fft_normalization, self.spatial_dims,
                                  hybrid=True)
        else:
            eta = fft.ifft2(kspace_prediction, **kw)
            eta = utils.coil_combination(eta, sensitivity_maps, method=self.coil_combination_method, dim=self.coil_dim)
        eta = torch.view_as_complex(eta)
        _, eta = utils.center_crop_to_smallest(target, eta)
        return eta

    forward_step = forward
