"""Drop-ins for `mridc.collections.reconstruction.models.recurrentvarnet.recurrentvarnet` (reference recurrentvarnet.py:17-240),
inference path."""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from mridc_amd import ops
from mridc_amd.collections.reconstruction.models.recurrentvarnet import conv2gru


class RecurrentInit(nn.Module):
    """Learned initialiser of the hidden state (recurrentvarnet.py:17-108): dilated replicate-padded 3x3 convs + ReLU, then one
    1x1 conv + ReLU per recurrent layer (every conv is one mrx_conv2d launch with the ReLU as its epilogue)."""

    def __init__(self, in_channels: int, out_channels: int, channels: Tuple[int, ...], dilations: Tuple[int, ...], depth: int = 2,
                 multiscale_depth: int = 1):
        super().__init__()
        self.conv_blocks = nn.ModuleList()
        self.out_blocks = nn.ModuleList()
        self.depth = depth
        self.multiscale_depth = multiscale_depth
        tch = in_channels
        for (curr_channels, curr_dilations) in zip(channels, dilations):
            self.conv_blocks.append(nn.Sequential(nn.ReplicationPad2d(curr_dilations),
                                                  nn.Conv2d(tch, curr_channels, 3, padding=0, dilation=curr_dilations)))
            tch = curr_channels
        tch = int(np.sum(channels[-multiscale_depth:]))
        for _ in range(depth):
            self.out_blocks.append(nn.Sequential(nn.Conv2d(tch, out_channels, 1, padding=0)))

    def forward(self, x: torch.Tensor, as_list: bool = False):
        features = []
        for block in self.conv_blocks:
            conv = block[-1]
            x = ops.conv2d(x, conv.weight, conv.bias, conv.dilation[0], ops.PAD_REPLICATE, ops.ACT_RELU)
            if self.multiscale_depth > 1:
                features.append(x)
        if self.multiscale_depth > 1:
            x = torch.cat(features[-self.multiscale_depth:], dim=1)
        outs = [ops.conv2d(x, block[-1].weight, block[-1].bias, 1, ops.PAD_ZERO, ops.ACT_RELU) for block in self.out_blocks]
        return outs if as_list else torch.stack(outs, dim=-1)


class RecurrentVarNetBlock(nn.Module):
    """One step of the Recurrent Variational Network (recurrentvarnet.py:111-240):
    k_{t+1} = k_t - alpha * where(mask == 0, 0, k_t - y) + F(S * H(sum_c conj(S) F^-1 k_t, h_t))
    = mrx_sens_reduce -> Conv2dGRU -> mrx_sens_expand -> mrx_dc_combine."""

    def __init__(self, in_channels: int = 2, hidden_channels: int = 64, num_layers: int = 4, fft_centered: bool = True,
                 fft_normalization: str = "ortho", spatial_dims: Optional[Tuple[int, int]] = None, coil_dim: int = 1):
        super().__init__()
        if in_channels != 2:
            raise NotImplementedError("mridc_amd RecurrentVarNetBlock: in_channels = 2 (one complex image) only")
        if coil_dim != 1:
            raise NotImplementedError("the HIP path expects the coil dimension at index 1")
        self.fft_centered = fft_centered
        self.fft_normalization = fft_normalization
        self.spatial_dims = spatial_dims if spatial_dims is not None else [-2, -1]
        self.coil_dim = coil_dim
        self.learning_rate = nn.Parameter(torch.tensor([1.0]))
        self._hybrid = False   # set by RecurrentVarNet for row-invariant masks: k-space arguments are IFFT_H(k), row transforms only
        self.regularizer = conv2gru.Conv2dGRU(in_channels=in_channels, hidden_channels=hidden_channels, num_layers=num_layers,
                                              replication_padding=True)

    def forward(self, current_kspace: torch.Tensor, masked_kspace: torch.Tensor, sampling_mask: torch.Tensor,
                sensitivity_map: torch.Tensor, hidden_state: Union[None, torch.Tensor, List[torch.Tensor]]):
        # hybrid steps chained by the model (W = 372): `_reduced_in` replaces this step's own sens_reduce, `_reduced_out` is the next one's
        if self._hybrid and getattr(self, "_reduced_in", None) is not None:
            img = self._reduced_in
        else:
            img = ops.sens_reduce(current_kspace, sensitivity_map, self.fft_centered, self.fft_normalization, self.spatial_dims,
                                  hybrid=self._hybrid)
        self._reduced_out = None
        term, hidden_state = self.regularizer(img.permute(0, 3, 1, 2), hidden_state, _complex_last=True)   # [B,H,W,2]
        # the update needs "+ F(S w)": the transform is linear, so expand -w and let the data-consistency kernel subtract it
        # (k - alpha * err - (-t) rounds exactly like k - alpha * err + t)
        neg = ops.scale(term, -1.0)
        if self._hybrid:                         # expand + the k-space update in one pass over the coil stack (the formula of dc_combine below)
            chain = getattr(self, "_want_reduced", False) and ops.sens_expand_dc_reduce_supported(sensitivity_map)
            res = ops.sens_expand_dc_hybrid(neg, sensitivity_map, current_kspace, masked_kspace, sampling_mask != 0, self.learning_rate,
                                            self.fft_centered, self.fft_normalization, reduce=chain)
            if chain:
                res, self._reduced_out = res
            return res, hidden_state
        minus_term = ops.sens_expand(neg, sensitivity_map, self.fft_centered, self.fft_normalization, self.spatial_dims,
                                     hybrid=self._hybrid)
        new_kspace = ops.dc_combine(current_kspace, current_kspace, masked_kspace, sampling_mask != 0, self.learning_rate, minus_term)
        return new_kspace, hidden_state
