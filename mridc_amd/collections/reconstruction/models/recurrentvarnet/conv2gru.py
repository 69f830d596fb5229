"""Drop-in for `mridc.collections.reconstruction.models.recurrentvarnet.conv2gru.Conv2dGRU` (reference conv2gru.py:10-163),
inference path.

Per layer: conv (5x5 / 3x3 dilation 2 / 3x3, replicate or zero padding) + ReLU is one launch (mrx_conv2d; the 3x3 layers into 64
features take its Winograd form, mrx_conv3x3_wino); the GRU on
cat(input, state) is ONE launch for the Recurrent VarNet's shape (1x1 gates, 64 features: mrx_conv2dgru_cell_1x1) and
conv + mrx_mul_sigmoid + conv + mrx_gru_blend otherwise.  `previous_state` is the reference's [B,hidden,H,W,layers] tensor;
a list of per-layer contiguous tensors is accepted as well (and then returned), which is what the RecurrentVarNet drop-in
passes between its blocks so the layer axis is never strided."""
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from mridc_amd import ops


class Conv2dGRU(nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: Optional[int] = None, num_layers: int = 2,
                 gru_kernel_size=1, orthogonal_initialization: bool = True, instance_norm: bool = False, dense_connect: int = 0,
                 replication_padding: bool = True):
        super().__init__()
        if out_channels is None:
            out_channels = in_channels
        if instance_norm or dense_connect:
            raise NotImplementedError("mridc_amd Conv2dGRU: instance_norm / dense_connect are not on the HIP path "
                                      "(the RecurrentVarNetBlock uses neither, recurrentvarnet.py:151-156)")
        self.num_layers = num_layers
        self.hidden_channels = hidden_channels
        self.dense_connect = dense_connect
        self.replication_padding = replication_padding
        self.gru_kernel_size = gru_kernel_size
        self.reset_gates = nn.ModuleList([])
        self.update_gates = nn.ModuleList([])
        self.out_gates = nn.ModuleList([])
        self.conv_blocks = nn.ModuleList([])
        for idx in range(num_layers + 1):                              # conv2gru.py:60-83
            in_ch = in_channels if idx == 0 else hidden_channels
            out_ch = hidden_channels if idx < num_layers else out_channels
            padding = 0 if replication_padding else (2 if idx == 0 else 1)
            block = []
            if replication_padding:
                block.append(nn.ReplicationPad2d(2 if idx in (0, 1) else 1))
            block.append(nn.Conv2d(in_channels=in_ch, out_channels=out_ch, kernel_size=5 if idx == 0 else 3,
                                   dilation=(2 if idx == 1 else 1), padding=padding))
            self.conv_blocks.append(nn.Sequential(*block))
        for _ in range(num_layers):                                    # conv2gru.py:85-99
            for gru_part in [self.reset_gates, self.update_gates, self.out_gates]:
                gru_part.append(nn.Sequential(nn.Conv2d(in_channels=2 * hidden_channels, out_channels=hidden_channels,
                                                        kernel_size=gru_kernel_size, padding=gru_kernel_size // 2)))
        if orthogonal_initialization:                                  # conv2gru.py:101-108
            for reset_gate, update_gate, out_gate in zip(self.reset_gates, self.update_gates, self.out_gates):
                nn.init.orthogonal_(reset_gate[-1].weight)
                nn.init.orthogonal_(update_gate[-1].weight)
                nn.init.orthogonal_(out_gate[-1].weight)
                nn.init.constant_(reset_gate[-1].bias, -1.0)
                nn.init.constant_(update_gate[-1].bias, 0.0)
                nn.init.constant_(out_gate[-1].bias, 0.0)
        self._pack_cache = {}

    def _conv(self, idx, x, relu):
        conv = self.conv_blocks[idx][-1]
        if not self.replication_padding and idx == 1:
            raise NotImplementedError("mridc_amd Conv2dGRU: the reference's zero-padded dilated layer (padding 1, dilation 2) shrinks "
                                      "the image; only replication_padding=True is on the HIP path")
        return ops.conv2d(x, conv.weight, conv.bias, conv.dilation[0], ops.PAD_REPLICATE if self.replication_padding else ops.PAD_ZERO,
                          ops.ACT_RELU if relu else ops.ACT_NONE)

    def _packed(self, idx):
        wu, wr, wo = (g[idx][-1].weight for g in (self.update_gates, self.reset_gates, self.out_gates))
        bs = [g[idx][-1].bias for g in (self.update_gates, self.reset_gates, self.out_gates)]
        key = tuple((t.data_ptr(), t._version) for t in (wu, wr, wo, *bs)) + (str(wu.device),)
        hit = self._pack_cache.get(idx)
        if hit is None or hit[0] != key:
            hit = (key, ops.conv2dgru_pack(wu, wr, wo), torch.cat([b.detach().reshape(-1) for b in bs]).contiguous())
            self._pack_cache[idx] = hit
        return hit[1], hit[2]

    def forward(self, cell_input: torch.Tensor, previous_state: Union[None, torch.Tensor, List[torch.Tensor]],
                _complex_last: bool = False) -> Tuple[torch.Tensor, Union[torch.Tensor, List[torch.Tensor]]]:
        """conv2gru.py:112-163.  `_complex_last` (used by RecurrentVarNetBlock): the output as [B,H,W,2] instead of [B,2,H,W]."""
        as_list = isinstance(previous_state, (list, tuple))
        if previous_state is None:
            states = [None] * self.num_layers                          # zeros (conv2gru.py:134-137) without materialising them
        elif as_list:
            states = list(previous_state)
        else:
            states = [previous_state[..., i].contiguous() for i in range(self.num_layers)]
        hid = self.hidden_channels
        fused = ops.conv2dgru_supported(hid, hid, self.gru_kernel_size)
        new_states = []
        x = cell_input
        for idx in range(self.num_layers):
            x = self._conv(idx, x, True)
            h = states[idx]
            if fused:
                packed, bias = self._packed(idx)
                new, x = ops.conv2dgru_cell_1x1(x, h, packed, bias, True)
            else:
                hz = h if h is not None else torch.zeros_like(x)
                stacked = torch.cat([x, hz], dim=1)
                ug, rg, og = self.update_gates[idx][-1], self.reset_gates[idx][-1], self.out_gates[idx][-1]
                pre_u = ops.conv2d(stacked, ug.weight, ug.bias, 1, ops.PAD_ZERO)
                pre_r = ops.conv2d(stacked, rg.weight, rg.bias, 1, ops.PAD_ZERO)
                pre_o = ops.conv2d(torch.cat([x, ops.mul_sigmoid(h, pre_r)], dim=1), og.weight, og.bias, 1, ops.PAD_ZERO)
                new, x = ops.gru_blend(h, pre_u, pre_o, True)
            new_states.append(new)
        last = self.conv_blocks[self.num_layers][-1]
        if _complex_last and last.out_channels == 2 and self.replication_padding:
            out = ops.conv_to_complex(x, last.weight, last.bias, last.dilation[0], ops.PAD_REPLICATE)
        else:
            out = self._conv(self.num_layers, x, False)
            if _complex_last:
                out = out.permute(0, 2, 3, 1)
        return out, (new_states if as_list else torch.stack(new_states, dim=-1))
