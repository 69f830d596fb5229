"""Drop-in for `mridc.collections.reconstruction.models.unet_base.unet_block` (reference unet_block.py:11-308).

Parameter containers are the same torch modules as the reference (identical state_dict keys and default init); the
forward pass runs on the HIP kernels: MFMA conv (csrc/conv.hip), InstanceNorm+LeakyReLU, pooling, pad, transpose conv
(csrc/unet.hip).
"""
import math
import os
from typing import List, Tuple

import torch

from mridc_amd import diff, ops


def _ns(x, *params, training=True):
    """`diff` (differentiable forms, training) when gradients are being recorded through x or the parameters, else `ops`."""
    return diff if diff.active(x, *params, training=training) else ops


class ConvBlock(torch.nn.Module):
    """unet_block.py:230-271: [Conv3x3 no bias -> InstanceNorm -> LeakyReLU(0.2) -> Dropout2d] x 2."""

    def __init__(self, in_chans: int, out_chans: int, drop_prob: float):
        super().__init__()
        self.in_chans = in_chans
        self.out_chans = out_chans
        self.drop_prob = drop_prob
        self.layers = torch.nn.Sequential(
            torch.nn.Conv2d(in_chans, out_chans, kernel_size=3, padding=1, bias=False),
            torch.nn.InstanceNorm2d(out_chans),
            torch.nn.LeakyReLU(negative_slope=0.2, inplace=True),
            torch.nn.Dropout2d(drop_prob),
            torch.nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
            torch.nn.InstanceNorm2d(out_chans),
            torch.nn.LeakyReLU(negative_slope=0.2, inplace=True),
            torch.nn.Dropout2d(drop_prob),
        )

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        if self.training and self.drop_prob > 0:
            raise NotImplementedError("Dropout2d in training mode is not part of the HIP inference path")
        o = _ns(image, self.layers[0].weight, self.layers[4].weight, training=self.training)
        x = o.conv_instance_norm_act(image, self.layers[0].weight, self.layers[1].eps, ops.ACT_LEAKY, 0.2)
        return o.conv_instance_norm_act(x, self.layers[4].weight, self.layers[5].eps, ops.ACT_LEAKY, 0.2)


class TransposeConvBlock(torch.nn.Module):
    """unet_block.py:274-308."""

    def __init__(self, in_chans: int, out_chans: int):
        super().__init__()
        self.in_chans = in_chans
        self.out_chans = out_chans
        self.layers = torch.nn.Sequential(
            torch.nn.ConvTranspose2d(in_chans, out_chans, kernel_size=2, stride=2, bias=False),
            torch.nn.InstanceNorm2d(out_chans),
            torch.nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        if diff.active(image, self.layers[0].weight, training=self.training):
            return diff.instance_norm_act(diff.conv_transpose2x2(image, self.layers[0].weight), self.layers[1].eps, ops.ACT_LEAKY, 0.2)
        return ops.conv_transpose2x2_instance_norm_act(image, self.layers[0].weight, self.layers[1].eps, ops.ACT_LEAKY, 0.2)


class Unet(torch.nn.Module):
    """unet_block.py:139-227."""

    def __init__(self, in_chans: int, out_chans: int, chans: int = 32, num_pool_layers: int = 4, drop_prob: float = 0.0):
        super().__init__()
        self.in_chans = in_chans
        self.out_chans = out_chans
        self.chans = chans
        self.num_pool_layers = num_pool_layers
        self.drop_prob = drop_prob
        self.down_sample_layers = torch.nn.ModuleList([ConvBlock(in_chans, chans, drop_prob)])
        ch = chans
        for _ in range(num_pool_layers - 1):
            self.down_sample_layers.append(ConvBlock(ch, ch * 2, drop_prob))
            ch *= 2
        self.conv = ConvBlock(ch, ch * 2, drop_prob)
        self.up_conv = torch.nn.ModuleList()
        self.up_transpose_conv = torch.nn.ModuleList()
        for _ in range(num_pool_layers - 1):
            self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch))
            self.up_conv.append(ConvBlock(ch * 2, ch, drop_prob))
            ch //= 2
        self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch))
        self.up_conv.append(torch.nn.Sequential(ConvBlock(ch * 2, ch, drop_prob),
                                                torch.nn.Conv2d(ch, self.out_chans, kernel_size=1, stride=1)))

    # the inference path keeps every Conv -> InstanceNorm -> LeakyReLU output as (raw, per-plane statistics) and lets the consumer normalise
    # while it loads: no apply pass, no concatenated tensor (csrc/unet_fused.hip); `fused = False` selects conv + apply launches
    fused = True                    # (class attribute: a test hook for the unfused formulation)

    def _fusable(self) -> bool:
        blocks = list(self.down_sample_layers) + [self.conv] + [c[0] if isinstance(c, torch.nn.Sequential) else c for c in self.up_conv]
        last = self.up_conv[-1]
        return (all(isinstance(b, ConvBlock) and b.layers[1].eps == b.layers[5].eps == 1e-5 and b.layers[2].negative_slope == 0.2
                    for b in blocks)
                and all(t.layers[1].eps == 1e-5 and t.layers[2].negative_slope == 0.2
                        and ops.unet_conv_transpose2x2_supported(t.in_chans, t.out_chans) for t in self.up_transpose_conv)
                and isinstance(last, torch.nn.Sequential) and last[1].out_channels <= 4 and last[1].in_channels <= 1024)

    def _forward_fused(self, image: torch.Tensor, tail=None) -> torch.Tensor:
        """unet_block.py:192-227 on (raw, statistics) pairs.  `tail(lazy, conv1x1 module)`: replaces the closing 1x1 convolution (NormUnet
        writes its un-normalised complex-last result from there)."""
        def block(b, a, skip=None):
            return ops.unet_conv3x3(ops.unet_conv3x3(a, skip, b.layers[0].weight), None, b.layers[4].weight)

        stack = []
        x = image
        for layer in self.down_sample_layers:                        # unet_block.py:203-206
            x = block(layer, x)
            stack.append(x)
            x = ops.unet_avg_pool2x2(x)
        x = block(self.conv, x)
        for transpose_conv, conv in zip(self.up_transpose_conv, self.up_conv):
            skip = stack.pop()
            x = ops.unet_conv_transpose2x2(x, transpose_conv.layers[0].weight)
            pad_r = 1 if x[0].shape[-1] != skip[0].shape[-1] else 0  # unet_block.py:215-222 (odd sizes only: written out, then padded)
            pad_b = 1 if x[0].shape[-2] != skip[0].shape[-2] else 0
            if pad_r or pad_b:
                x = ops.pad2d(ops.unet_apply(x), 0, pad_b, 0, pad_r, mode=1)
            last = isinstance(conv, torch.nn.Sequential)
            x = block(conv[0] if last else conv, x, skip)             # torch.cat([output, downsample_layer], dim=1) read in place
            if last:
                return tail(x, conv[1]) if tail is not None else ops.unet_conv1x1(x, conv[1].weight, conv[1].bias)
        return ops.unet_apply(x)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        stack = []
        output = image
        o = _ns(image, *self.parameters(), training=self.training)
        if o is ops and self.fused and not self.training and self._fusable():
            return self._forward_fused(image)
        for layer in self.down_sample_layers:                        # unet_block.py:203-206
            output = layer(output)
            stack.append(output)
            output = o.avg_pool2x2(output)
        output = self.conv(output)
        for transpose_conv, conv in zip(self.up_transpose_conv, self.up_conv):
            downsample_layer = stack.pop()
            output = transpose_conv(output)
            pad_r = 1 if output.shape[-1] != downsample_layer.shape[-1] else 0     # unet_block.py:215-222
            pad_b = 1 if output.shape[-2] != downsample_layer.shape[-2] else 0
            if pad_r or pad_b:
                output = o.pad2d(output, 0, pad_b, 0, pad_r, mode=1)
            output = o.concat_channels(output, downsample_layer)
            if isinstance(conv, torch.nn.Sequential):
                output = conv[0](output)
                output = o.conv2d(output, conv[1].weight, conv[1].bias, 1, ops.PAD_ZERO)
            else:
                output = conv(output)
        return output


class NormUnet(torch.nn.Module):
    """unet_block.py:11-136."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0,
                 padding_size: int = 15, normalize: bool = True, norm_groups: int = 2):
        super().__init__()
        self.unet = Unet(in_chans=in_chans, out_chans=out_chans, chans=chans, num_pool_layers=num_pools,
                         drop_prob=drop_prob)
        self.padding_size = padding_size
        self.normalize = normalize
        self.norm_groups = norm_groups

    @staticmethod
    def complex_to_chan_dim(x: torch.Tensor) -> torch.Tensor:
        b, c, h, w, two = x.shape
        if two != 2:
            raise AssertionError
        return x.permute(0, 4, 1, 2, 3).reshape(b, 2 * c, h, w)

    @staticmethod
    def chan_complex_to_last_dim(x: torch.Tensor) -> torch.Tensor:
        b, c2, h, w = x.shape
        if c2 % 2 != 0:
            raise AssertionError
        c = c2 // 2
        return x.view(b, 2, c, h, w).permute(0, 2, 3, 4, 1).contiguous()

    def norm(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return _ns(x).group_norm(x, self.norm_groups)

    def unnorm(self, x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
        return _ns(x, mean).group_unnorm(x, mean, std, self.norm_groups)

    def pad(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[List[int], List[int], int, int]]:
        _, _, h, w = x.shape
        w_mult = ((w - 1) | self.padding_size) + 1
        h_mult = ((h - 1) | self.padding_size) + 1
        w_pad = [math.floor((w_mult - w) / 2), math.ceil((w_mult - w) / 2)]
        h_pad = [math.floor((h_mult - h) / 2), math.ceil((h_mult - h) / 2)]
        x = _ns(x).pad2d(x, h_pad[0], h_pad[1], w_pad[0], w_pad[1], mode=0)
        return x, (h_pad, w_pad, h_mult, w_mult)

    @staticmethod
    def unpad(x: torch.Tensor, h_pad: List[int], w_pad: List[int], h_mult: int, w_mult: int) -> torch.Tensor:
        return _ns(x).pad2d(x, -h_pad[0], -h_pad[1], -w_pad[0], -w_pad[1], mode=0)

    def _fused_ok(self, x: torch.Tensor) -> bool:
        """complex-last input with one or two coils, two norm groups, the inference path of a fusable U-Net: head and tail each in one pass."""
        u = self.unet
        return (x.dim() == 5 and x.shape[-1] == 2 and x.shape[1] <= 2 and self.normalize and self.norm_groups == 2 and Unet.fused
                and not u.training and not diff.active(x, *u.parameters()) and u.in_chans == u.out_chans == 2 * x.shape[1] and u._fusable())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fused_ok(x):
            _, _, h, w, _ = x.shape
            w_mult = ((w - 1) | self.padding_size) + 1                        # unet_block.py:93-112
            h_mult = ((h - 1) | self.padding_size) + 1
            w_pad = [math.floor((w_mult - w) / 2), math.ceil((w_mult - w) / 2)]
            h_pad = [math.floor((h_mult - h) / 2), math.ceil((h_mult - h) / 2)]
            xp, mean, std = ops.unet_cnorm_pad(x, h_pad, w_pad)
            return self.unet._forward_fused(xp, tail=lambda lazy, conv: ops.unet_conv1x1_cunnorm(lazy, conv.weight, conv.bias, mean, std,
                                                                                                 h_pad[0], w_pad[0], h, w))
        iscomplex = False
        if x.shape[-1] == 2:
            x = self.complex_to_chan_dim(x)
            iscomplex = True
        mean = std = None
        if self.normalize:
            x, mean, std = self.norm(x)
        x, pad_sizes = self.pad(x)
        x = self.unet(x)
        x = self.unpad(x, *pad_sizes)
        if self.normalize:
            x = self.unnorm(x, mean, std)
        if iscomplex:
            x = self.chan_complex_to_last_dim(x)
        return x
