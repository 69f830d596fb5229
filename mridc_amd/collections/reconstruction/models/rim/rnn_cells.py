"""Drop-in for `mridc.collections.reconstruction.models.rim.rnn_cells` (reference rnn_cells.py:8-391), 2-D only."""
import torch
import torch.nn as nn

from mridc_amd import diff, ops


def _orthogonalize(weights, chunks=1):
    return torch.cat([nn.init.orthogonal_(w) for w in weights.chunk(chunks, 0)], 0)


class _GatedCellBase(nn.Module):
    GATES = 3

    def __init__(self, input_size, hidden_size, conv_dim, kernel_size, dilation, bias):
        super().__init__()
        if conv_dim != 2:
            raise NotImplementedError("mridc_amd implements the 2-D convolutional path (conv_dim=2) only")
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.bias = bias
        self.conv_dim = conv_dim
        pad = int(dilation * (kernel_size - 1) / 2)
        self.ih = nn.Conv2d(input_size, self.GATES * hidden_size, kernel_size, padding=pad, dilation=dilation, bias=bias)
        self.hh = nn.Conv2d(hidden_size, self.GATES * hidden_size, kernel_size, padding=pad, dilation=dilation, bias=False)
        self.reset_parameters()

    def check_forward_input(self, _input):
        if _input.size(1) != self.input_size:
            raise RuntimeError(f"input has inconsistent input_size: got {_input.size(1)}, expected {self.input_size}")

    def check_forward_hidden(self, _input, hx, hidden_label=""):
        if _input.size(0) != hx.size(0):
            raise RuntimeError(f"Input batch size {_input.size(0)} doesn't match hidden{hidden_label} batch size {hx.size(0)}")
        if hx.size(1) != self.hidden_size:
            raise RuntimeError(f"hidden{hidden_label} has inconsistent hidden_size: got {hx.size(1)}, expected {self.hidden_size}")

    def _ns(self, _input, hx):
        """`diff` (training: convolutions with HIP backward kernels, gates recorded by torch) or `ops`."""
        return diff if diff.active(_input, hx, *self.parameters(), training=self.training) else ops

    def _convs(self, _input, hx):
        o = self._ns(_input, hx)
        ih = o.conv2d(_input, self.ih.weight, self.ih.bias, self.dilation, ops.PAD_ZERO)
        hh = o.conv2d(hx, self.hh.weight, None, self.dilation, ops.PAD_ZERO)
        return ih, hh


class ConvGRUCell(_GatedCellBase):
    """rnn_cells.py:8-127."""
    GATES = 3

    def __init__(self, input_size, hidden_size, conv_dim, kernel_size, dilation=1, bias=True):
        super().__init__(input_size, hidden_size, conv_dim, kernel_size, dilation, bias)

    def reset_parameters(self):
        self.ih.weight.data = _orthogonalize(self.ih.weight.data)
        self.hh.weight.data = _orthogonalize(self.hh.weight.data)
        if self.bias is True:
            nn.init.zeros_(self.ih.bias)

    def forward(self, _input, hx):
        ih, hh = self._convs(_input, hx)
        return self._ns(_input, hx).gru_gates(ih, hh, hx)          # rnn_cells.py:118-127


class ConvMGUCell(_GatedCellBase):
    """rnn_cells.py:130-261."""
    GATES = 2

    def __init__(self, input_size, hidden_size, conv_dim, kernel_size, dilation=1, bias=True):
        super().__init__(input_size, hidden_size, conv_dim, kernel_size, dilation, bias)

    def reset_parameters(self):
        self.ih.weight.data = _orthogonalize(self.ih.weight.data)
        self.hh.weight.data = _orthogonalize(self.hh.weight.data)
        nn.init.xavier_uniform_(self.ih.weight, nn.init.calculate_gain("relu"))
        nn.init.xavier_uniform_(self.hh.weight)
        if self.bias is True:
            nn.init.zeros_(self.ih.bias)

    def forward(self, _input, hx):
        ih, hh = self._convs(_input, hx)
        return self._ns(_input, hx).mgu_gates(ih, hh, hx)          # rnn_cells.py:255-261


class IndRNNCell(nn.Module):
    """rnn_cells.py:264-391: ReLU(ih(x) + hh * hx), `ih` a zero-padded conv, `hh` a per-channel weight."""

    def __init__(self, input_size, hidden_size, conv_dim, kernel_size, dilation=1, bias=True):
        super().__init__()
        if conv_dim not in (2, 3):
            raise NotImplementedError("mridc_amd implements conv_dim = 2 (HIP kernels) and conv_dim = 3 (torch device ops)")
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.bias = bias
        self.conv_dim = conv_dim
        self.ih = (nn.Conv2d if conv_dim == 2 else nn.Conv3d)(input_size, hidden_size, kernel_size,
                                                              padding=int(dilation * (kernel_size - 1) / 2), dilation=dilation, bias=bias)
        self.hh = nn.Parameter(nn.init.normal_(torch.empty(*([1, hidden_size] + [1] * conv_dim)),
                                               std=1.0 / (hidden_size * (1 + kernel_size ** 2))))
        self.reset_parameters()

    def reset_parameters(self):
        """rnn_cells.py:315-322."""
        self.ih.weight.data = _orthogonalize(self.ih.weight.data)
        nn.init.normal_(self.ih.weight, std=1.0 / (self.hidden_size * (1 + self.kernel_size ** 2)))
        if self.bias is True:
            nn.init.zeros_(self.ih.bias)

    def check_forward_input(self, _input):
        if _input.size(1) != self.input_size:
            raise RuntimeError(f"input has inconsistent input_size: got {_input.size(1)}, expected {self.input_size}")

    def forward(self, _input, hx):
        if self.conv_dim == 3:                                        # rnn_cells.py:384-391 (torch device ops: outside the HIP hot path)
            _input = _input.unsqueeze(0)
            hx = hx.permute(1, 0, 2, 3).unsqueeze(0)
            return torch.relu(self.ih(_input) + self.hh * hx)
        return ops.indrnn_cell(_input, self.ih.weight, self.ih.bias, self.hh, hx, self.dilation)
