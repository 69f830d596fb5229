"""Drop-in for `mridc.collections.reconstruction.models.rim.conv_layers` (reference conv_layers.py:8-123)."""
import torch
import torch.nn as nn

from mridc_amd import ops


class ConvRNNStack(nn.Module):
    """conv_layers.py:8-33."""

    def __init__(self, convs, rnn):
        super().__init__()
        self.convs = convs
        self.rnn = rnn

    def forward(self, x, hidden):
        return self.rnn(self.convs(x), hidden)


class ConvNonlinear(nn.Module):
    """conv_layers.py:36-123: ReplicationPad(dil*(k-1)//2) -> Conv(padding=0, dilation) -> ReLU / LeakyReLU / identity.

    `conv_layer` is an nn.Conv2d used as the parameter container (same state_dict keys and the same
    initialisation as the reference); the arithmetic is mrx_conv2d with the replicate border folded into the tile loader.
    conv_dim = 3 (the reference's 3-D mode, rim_block.py:168-180) keeps an nn.Conv3d and runs as a composition of torch device ops
    (ReplicationPad3d + conv3d): it is outside the hand-written hot path, kept so that 3-D configs load and run.
    """

    def __init__(self, input_size, features, conv_dim, kernel_size, dilation, bias, nonlinear="relu"):
        super().__init__()
        if conv_dim not in (2, 3):
            raise NotImplementedError("mridc_amd implements conv_dim = 2 (HIP kernels) and conv_dim = 3 (torch device ops)")
        self.input_size = input_size
        self.features = features
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.bias = bias
        self.conv_dim = conv_dim
        if nonlinear is not None and nonlinear.upper() == "RELU":
            self.act, self.slope = ops.ACT_RELU, 0.0
        elif nonlinear is not None and nonlinear.upper() == "LEAKYRELU":
            self.act, self.slope = ops.ACT_LEAKY, 0.01      # torch.nn.LeakyReLU() default (conv_layers.py:66)
        elif nonlinear is None:
            self.act, self.slope = ops.ACT_NONE, 0.0
        else:
            raise ValueError("Please specify a proper nonlinearity")
        self.conv_layer = (nn.Conv2d if conv_dim == 2 else nn.Conv3d)(in_channels=input_size, out_channels=features, kernel_size=kernel_size,
                                                                      padding=0, dilation=dilation, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        """conv_layers.py:89-94."""
        torch.nn.init.kaiming_normal_(self.conv_layer.weight, nonlinearity="relu")
        if self.conv_layer.bias is not None:
            nn.init.zeros_(self.conv_layer.bias)

    def check_forward_input(self, _input):
        if _input.size(1) != self.input_size:
            raise RuntimeError(f"input has inconsistent input_size: got {_input.size(1)}, expected {self.input_size}")

    def forward(self, _input):
        if self.conv_dim == 3:                                        # conv_layers.py:72-85,121-123 on [C, D, H, W] / [N, C, D, H, W]
            p = int(self.dilation * (self.kernel_size - 1) / 2)
            x = _input.unsqueeze(0) if _input.dim() == 4 else _input
            x = torch.nn.functional.pad(x, (p, p, p, p, p, p), mode="replicate") if p > 0 else x
            x = torch.nn.functional.conv3d(x, self.conv_layer.weight, self.conv_layer.bias, padding=0, dilation=self.dilation)
            x = x.squeeze(0) if _input.dim() == 4 else x
            if self.act == ops.ACT_RELU:
                return torch.relu(x)
            return torch.nn.functional.leaky_relu(x, self.slope) if self.act == ops.ACT_LEAKY else x
        return ops.conv2d(_input, self.conv_layer.weight, self.conv_layer.bias, self.dilation, ops.PAD_REPLICATE,
                          self.act, self.slope)
