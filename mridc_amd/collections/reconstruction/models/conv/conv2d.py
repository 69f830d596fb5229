"""Drop-in for `mridc.collections.reconstruction.models.conv.conv2d.Conv2d` (reference conv/conv2d.py:8-69), inference path.

A cascade of 3x3 convolutions (zero padding 1) with an optional BatchNorm2d (eps 1e-4) and an activation after every
convolution but the last.  Each (conv, BatchNorm in eval mode, activation) group is ONE mrx_conv2d launch: the BatchNorm
affine of the running statistics is folded into the conv weights and bias (a per-parameter-version host-side prep of a few
hundred floats), the activation is the kernel's epilogue.  Parameter names are the reference's (`conv.{i}.weight`, ...)."""
import torch
import torch.nn as nn

from mridc_amd import ops


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels, n_convs=3, activation=nn.PReLU(), batchnorm=False):
        super().__init__()
        conv = []
        for idx in range(n_convs):                                   # conv2d.py:34-49
            conv.append(nn.Conv2d(in_channels if idx == 0 else hidden_channels,
                                  hidden_channels if idx != n_convs - 1 else out_channels, kernel_size=3, padding=1))
            if batchnorm:
                conv.append(nn.BatchNorm2d(hidden_channels if idx != n_convs - 1 else out_channels, eps=1e-4))
            if idx != n_convs - 1:
                conv.append(activation)
        self.conv = nn.Sequential(*conv)
        self._cache = {}

    @staticmethod
    def _version(*tensors):
        return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)

    def _activation(self, mod):
        """(act code, slope) of an activation module; the PReLU slope is read back once per parameter version."""
        if isinstance(mod, nn.PReLU):
            if mod.weight.numel() != 1:
                raise NotImplementedError("mridc_amd Conv2d: per-channel PReLU is not on the HIP path")
            key = ("prelu", id(mod)) + self._version(mod.weight)
            if key not in self._cache:
                self._cache = {k: v for k, v in self._cache.items() if k[:2] != key[:2]}
                self._cache[key] = float(mod.weight.detach().reshape(-1)[0])
            return ops.ACT_LEAKY, self._cache[key]
        if isinstance(mod, nn.LeakyReLU):
            return ops.ACT_LEAKY, float(mod.negative_slope)
        if isinstance(mod, nn.ReLU):
            return ops.ACT_RELU, 0.0
        raise NotImplementedError(f"mridc_amd Conv2d: activation {type(mod).__name__} is not on the HIP path")

    def _folded(self, i, conv, bn):
        """Conv weights / bias with the eval-mode BatchNorm affine folded in: y = (conv(x) - mean) * gamma / sqrt(var + eps) + beta."""
        if bn.training:
            raise NotImplementedError("mridc_amd Conv2d: BatchNorm in training mode (batch statistics) is not on the HIP path")
        key = ("bn", i) + self._version(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        if key not in self._cache:
            self._cache = {k: v for k, v in self._cache.items() if k[:2] != key[:2]}
            with torch.no_grad():
                s = torch.rsqrt(bn.running_var + bn.eps)
                if bn.weight is not None:
                    s = s * bn.weight
                w = conv.weight * s.reshape(-1, 1, 1, 1)
                b = (conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)) - bn.running_mean
                b = b * s
                if bn.bias is not None:
                    b = b + bn.bias
            self._cache[key] = (w.contiguous(), b.contiguous())
        return self._cache[key]

    def forward(self, x, _complex_last=False):
        """conv2d.py:54-69.  `_complex_last` (used by the cascade blocks of this package): return permute(out, (0, 2, 3, 1)) --
        what every caller does next -- with the last convolution writing that layout directly (2 output channels)."""
        if x.dim() == 5:
            x = x.squeeze(1)
            if x.shape[-1] == 2:
                x = x.permute(0, 3, 1, 2)
        mods = list(self.conv)
        i = 0
        while i < len(mods):
            conv = mods[i]
            j = i + 1
            w, b = conv.weight, conv.bias
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
                w, b = self._folded(i, conv, mods[j])
                j += 1
            act, slope = ops.ACT_NONE, 0.0
            if j < len(mods) and not isinstance(mods[j], nn.Conv2d):
                act, slope = self._activation(mods[j])
                j += 1
            if _complex_last and j >= len(mods) and act == ops.ACT_NONE and conv.out_channels == 2:
                return ops.conv_to_complex(x, w, b, 1, ops.PAD_ZERO)
            x = ops.conv2d(x, w, b, 1, ops.PAD_ZERO, act, slope)
            i = j
        return x.permute(0, 2, 3, 1) if _complex_last else x
