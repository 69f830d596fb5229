"""Oracle: NormUnet / Unet (reference mridc/collections/reconstruction/models/unet_base/unet_block.py).

Test infrastructure.  Functional: weights come in a dict keyed like `NormUnet.state_dict()`
(`unet.down_sample_layers.{l}.layers.{0,4}.weight`, `unet.conv.layers.{0,4}.weight`,
`unet.up_transpose_conv.{l}.layers.0.weight`, `unet.up_conv.{l}.layers.{0,4}.weight`, last level
`unet.up_conv.{L}.0.layers.{0,4}.weight`, `unet.up_conv.{L}.1.weight/.bias`), SURVEY appendix B.
"""
import math

import torch
import torch.nn.functional as F


_CONV3X3 = [F.conv2d]      # the 3x3 convolutions of the blocks (unet_block.py:250-259): oracle.amp.fp16_kernel_arithmetic swaps in the precision-16 kernels' arithmetic


def _in_lrelu(x):
    # InstanceNorm2d (no affine, eps 1e-5, biased variance) + LeakyReLU(0.2); unet_block.py:252-253
    return F.leaky_relu(F.instance_norm(x, eps=1e-5), negative_slope=0.2)


def conv_block(x, w0, w1):
    """unet_block.py:250-259 (Dropout2d(p=0) is the identity at eval)."""
    x = _in_lrelu(_CONV3X3[0](x, w0, None, padding=1))
    return _in_lrelu(_CONV3X3[0](x, w1, None, padding=1))


def transpose_conv_block(x, w):
    """unet_block.py:292-296."""
    return _in_lrelu(F.conv_transpose2d(x, w, None, stride=2))


def unet_forward(p, x, num_pool_layers, prefix="unet."):
    """unet_block.py:189-227."""
    stack = []
    out = x
    for l in range(num_pool_layers):
        pre = f"{prefix}down_sample_layers.{l}.layers."
        out = conv_block(out, p[pre + "0.weight"], p[pre + "4.weight"])
        stack.append(out)
        out = F.avg_pool2d(out, kernel_size=2, stride=2, padding=0)
    out = conv_block(out, p[prefix + "conv.layers.0.weight"], p[prefix + "conv.layers.4.weight"])
    for l in range(num_pool_layers):
        skip = stack.pop()
        out = transpose_conv_block(out, p[f"{prefix}up_transpose_conv.{l}.layers.0.weight"])
        pad = [0, 0, 0, 0]
        if out.shape[-1] != skip.shape[-1]:
            pad[1] = 1
        if out.shape[-2] != skip.shape[-2]:
            pad[3] = 1
        if sum(pad) != 0:
            out = F.pad(out, pad, "reflect")       # :215-222
        out = torch.cat([out, skip], dim=1)
        if l < num_pool_layers - 1:
            pre = f"{prefix}up_conv.{l}.layers."
            out = conv_block(out, p[pre + "0.weight"], p[pre + "4.weight"])
        else:
            pre = f"{prefix}up_conv.{l}.0.layers."
            out = conv_block(out, p[pre + "0.weight"], p[pre + "4.weight"])
            out = F.conv2d(out, p[f"{prefix}up_conv.{l}.1.weight"], p[f"{prefix}up_conv.{l}.1.bias"])
    return out


def norm_unet_forward(p, x, num_pools, padding_size=15, normalize=True, norm_groups=2, prefix="unet."):
    """unet_block.py:113-136.  x: [B,C,H,W,2] (complex last) or [B,C,H,W]."""
    iscomplex = x.shape[-1] == 2
    if iscomplex:                                   # complex_to_chan_dim :55-60
        b, c, h, w, _ = x.shape
        x = x.permute(0, 4, 1, 2, 3).reshape(b, 2 * c, h, w)
    b, c, h, w = x.shape
    mean = std = None
    if normalize:                                   # norm :71-85 (unbiased std, no eps)
        g = x.reshape(b, norm_groups, -1)
        mean = g.mean(-1, keepdim=True)
        std = g.std(-1, keepdim=True)
        x = ((g - mean) / std).reshape(b, c, h, w)
    w_mult = ((w - 1) | padding_size) + 1           # pad :93-106
    h_mult = ((h - 1) | padding_size) + 1
    w_pad = [math.floor((w_mult - w) / 2), math.ceil((w_mult - w) / 2)]
    h_pad = [math.floor((h_mult - h) / 2), math.ceil((h_mult - h) / 2)]
    x = F.pad(x, w_pad + h_pad)
    x = unet_forward(p, x, num_pools, prefix)
    x = x[..., h_pad[0]: h_mult - h_pad[1], w_pad[0]: w_mult - w_pad[1]]     # unpad :109-111
    if normalize:                                   # unnorm :87-91
        b2, c2, h2, w2 = x.shape
        x = (x.reshape(b2, norm_groups, -1) * std + mean).reshape(b2, c2, h2, w2)
    if iscomplex:                                   # chan_complex_to_last_dim :62-69
        b2, c2, h2, w2 = x.shape
        x = x.view(b2, 2, c2 // 2, h2, w2).permute(0, 2, 3, 4, 1).contiguous()
    return x
