"""Oracle: RIM cascade block (reference mridc/collections/reconstruction/models/rim/*).  Test infrastructure.

Functional restatement: weights are passed as a plain dict keyed exactly like the reference
`RIMBlock.state_dict()`, so golden fixtures can carry reference weights as data.
The 2-D mode (conv_dim == 2, dimensionality == 2) and the 3-D mode (conv_dim == dimensionality == 3, IndRNN layers) are restated.
"""
import torch
import torch.nn.functional as F

from . import fft as offt
from . import utils as outils


# The 2-D convolutions of the regulariser go through this hook so that oracle/amp.py can restate them with half-precision operands
# (the reference trains under AMP: base_cirim_train.yaml:180); the default is torch's fp32 convolution.
_CONV2D = [F.conv2d]


def _conv2d(x, weight, bias=None, padding=0, dilation=1):
    return _CONV2D[0](x, weight, bias, padding=padding, dilation=dilation)


# Hook on the hidden state a recurrent cell returns (identity by default): oracle/amp.py's restatement of the precision-16 KERNELS rounds it to fp16
# here (csrc/rim_amp16.hip keeps its states in fp16; torch.autocast itself keeps them fp32).
_STATE = [lambda h: h]


def log_likelihood_gradient(eta, masked_kspace, sense, mask, sigma, fft_centered, fft_normalization,
                            spatial_dims, coil_dim):
    """rim_utils.py:11-67.  eta [B,H,W,2]; y,S [B,C,H,W,2]; mask broadcastable -> [B,4,H,W]."""
    if coil_dim == 0:  # rim_utils.py:41-42
        coil_dim = 1
    e = eta.unsqueeze(coil_dim)                                   # [B,1,H,W,2]
    k = offt.fft2(outils.complex_mul(e, sense), fft_centered, fft_normalization, spatial_dims)   # :44-51
    r = offt.ifft2(mask * (k - masked_kspace), fft_centered, fft_normalization, spatial_dims)   # :53-58
    g = outils.complex_mul(r, outils.complex_conj(sense)).sum(coil_dim) / (sigma ** 2.0)         # :61-62
    return torch.cat((eta[..., 0:1], eta[..., 1:2], g[..., 0:1], g[..., 1:2]), -1).permute(0, 3, 1, 2)  # :67


def conv_nonlinear(x, weight, bias, kernel_size, dilation, nonlinear):
    """conv_layers.py:72-85,121-123: replication pad by dil*(k-1)//2, conv(padding=0), activation."""
    p = int(dilation * (kernel_size - 1) / 2)
    x = F.pad(x, (p, p, p, p), mode="replicate") if p > 0 else x
    x = _conv2d(x, weight, bias, padding=0, dilation=dilation)
    if nonlinear is None:
        return x
    if nonlinear.upper() == "RELU":
        return F.relu(x)
    if nonlinear.upper() == "LEAKYRELU":
        return F.leaky_relu(x)  # torch default slope 0.01, conv_layers.py:66
    raise ValueError("Please specify a proper nonlinearity")


def _zero_pad(kernel_size, dilation):
    return int(dilation * (kernel_size - 1) / 2)   # rnn_cells.py:27,161,299


def indrnn_cell(x, hx, ih_w, ih_b, hh, kernel_size, dilation):
    """rnn_cells.py:295-312,384-391: ReLU(conv_zero_pad(x) + hh * hx)."""
    p = _zero_pad(kernel_size, dilation)
    return _STATE[0](F.relu(_conv2d(x, ih_w, ih_b, padding=p, dilation=dilation) + hh * hx))


def convgru_cell(x, hx, ih_w, ih_b, hh_w, kernel_size, dilation):
    """rnn_cells.py:23-38,112-127."""
    p = _zero_pad(kernel_size, dilation)
    i_r, i_z, i_n = _conv2d(x, ih_w, ih_b, padding=p, dilation=dilation).chunk(3, 1)
    h_r, h_z, h_n = _conv2d(hx, hh_w, None, padding=p, dilation=dilation).chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return n * (1 - z) + z * hx


def convmgu_cell(x, hx, ih_w, ih_b, hh_w, kernel_size, dilation):
    """rnn_cells.py:157-172,249-261."""
    p = _zero_pad(kernel_size, dilation)
    i_f, i_c = _conv2d(x, ih_w, ih_b, padding=p, dilation=dilation).chunk(2, 1)
    h_f, h_c = _conv2d(hx, hh_w, None, padding=p, dilation=dilation).chunk(2, 1)
    f = torch.sigmoid(i_f + h_f)
    c = torch.tanh(i_c + f * h_c)
    return c + f * (hx - c)


class RIMConfig:
    """Hyper-parameters of one RIMBlock (rim_block.py:18-39), plain attributes."""

    def __init__(self, recurrent_layer="IndRNN", conv_filters=(64, 64, 2), conv_kernels=(5, 3, 3),
                 conv_dilations=(1, 2, 1), conv_bias=(True, True, False), recurrent_filters=(64, 64, 0),
                 recurrent_kernels=(1, 1, 0), recurrent_dilations=(1, 1, 0), recurrent_bias=(True, True, False),
                 depth=2, time_steps=8, conv_dim=2, no_dc=False, fft_centered=True, fft_normalization="ortho",
                 spatial_dims=None, coil_dim=1, dimensionality=2):
        if (conv_dim, dimensionality) not in ((2, 2), (3, 3)):
            raise NotImplementedError("oracle restates the 2-D mode and the 3-D mode (conv_dim = dimensionality = 3)")
        self.conv_dim, self.dimensionality = conv_dim, dimensionality
        self.recurrent_layer = recurrent_layer
        self.conv_filters, self.conv_kernels = list(conv_filters), list(conv_kernels)
        self.conv_dilations, self.conv_bias = list(conv_dilations), list(conv_bias)
        self.recurrent_filters, self.recurrent_kernels = list(recurrent_filters), list(recurrent_kernels)
        self.recurrent_dilations, self.recurrent_bias = list(recurrent_dilations), list(recurrent_bias)
        self.depth, self.time_steps, self.no_dc = depth, time_steps, no_dc
        self.input_size = None
        self.fft_centered, self.fft_normalization = fft_centered, fft_normalization
        self.spatial_dims = [-2, -1] if spatial_dims is None else list(spatial_dims)
        self.coil_dim = coil_dim

    def layer_table(self):
        """rim_block.py:70-121: (conv spec, rnn spec) per stacked layer; the last conv is `final_layer`."""
        nonlin = ["relu", "relu", None]
        rnn_t = [self.recurrent_layer, self.recurrent_layer, None]
        layers, final = [], None
        for i in range(len(self.conv_filters)):
            conv = dict(k=self.conv_kernels[i], d=self.conv_dilations[i], nl=nonlin[i]) if self.conv_filters[i] else None
            if self.recurrent_filters[i] != 0 and rnn_t[i] is not None:
                t = rnn_t[i].upper()
                if t not in ("GRU", "MGU", "INDRNN"):
                    raise ValueError("Please specify a proper recurrent layer type.")
                layers.append((conv, dict(type=t, k=self.recurrent_kernels[i], d=self.recurrent_dilations[i])))
            final = conv
        return layers, final


def _rnn_apply(p, pre, spec, x, h):
    if spec["type"] == "INDRNN":
        return indrnn_cell(x, h, p[pre + "ih.weight"], p.get(pre + "ih.bias"), p[pre + "hh"], spec["k"], spec["d"])
    if spec["type"] == "GRU":
        return convgru_cell(x, h, p[pre + "ih.weight"], p.get(pre + "ih.bias"), p[pre + "hh.weight"], spec["k"], spec["d"])
    return convmgu_cell(x, h, p[pre + "ih.weight"], p.get(pre + "ih.bias"), p[pre + "hh.weight"], spec["k"], spec["d"])


def _conv3d_nonlinear(x, weight, bias, kernel_size, dilation, nonlinear):
    """conv_layers.py:72-85,121-123 with conv_dim = 3 on an unbatched [C, D, H, W] input: ReplicationPad3d, Conv3d(padding=0), activation."""
    p = int(dilation * (kernel_size - 1) / 2)
    x = F.pad(x.unsqueeze(0), (p, p, p, p, p, p), mode="replicate") if p > 0 else x.unsqueeze(0)
    x = F.conv3d(x, weight, bias, padding=0, dilation=dilation).squeeze(0)
    return x if nonlinear is None else (F.relu(x) if nonlinear.upper() == "RELU" else F.leaky_relu(x))


def rim_block_forward_3d(p, cfg, pred, masked_kspace, sense, mask, eta=None, hx=None, sigma=1.0, keep_eta=False):
    """rim_block.py:168-180,217-249 with dimensionality = 3 (IndRNN layers): slices folded into the batch for the data-consistency
    gradient, the regulariser a 3-D convolution over (batch * slices, H, W).  Returns (list of etas [B*S,H,W,2], hx [B*S,f,H,W])."""
    batch, slices = masked_kspace.shape[0], masked_kspace.shape[1]
    fold = lambda t: t.reshape([t.shape[0] * t.shape[1], *t.shape[2:]])  # noqa: E731
    pred = pred[-1].detach() if isinstance(pred, (tuple, list)) else fold(pred)
    masked_kspace, mask, sense = fold(masked_kspace), fold(mask), fold(sense)
    if hx is None:
        hx = [masked_kspace.new_zeros((masked_kspace.size(0), f, *masked_kspace.size()[2:-1])) for f in cfg.recurrent_filters if f != 0]
    else:
        hx = list(hx)
    if eta is None or eta.ndim < 3:
        if keep_eta:
            eta = pred
        else:
            img = offt.ifft2(pred, cfg.fft_centered, cfg.fft_normalization, cfg.spatial_dims)
            eta = outils.complex_mul(img, outils.complex_conj(sense)).sum(cfg.coil_dim)
    if eta.dim() == 5:
        eta = fold(eta)
    layers, final = cfg.layer_table()
    etas = []
    for _ in range(cfg.time_steps):
        g = log_likelihood_gradient(eta, masked_kspace, sense, mask, sigma, cfg.fft_centered, cfg.fft_normalization, cfg.spatial_dims,
                                    cfg.coil_dim).contiguous()
        g = g.view([batch * slices, 4, g.shape[2], g.shape[3]]).permute(1, 0, 2, 3)                # :230-231 -> [4, D, H, W]
        for li, (conv, rnn) in enumerate(layers):
            pre = f"layers.{li}."
            g = _conv3d_nonlinear(g, p[pre + "convs.conv_layer.weight"], p.get(pre + "convs.conv_layer.bias"), conv["k"], conv["d"], conv["nl"])
            if rnn["type"] != "INDRNN":
                raise NotImplementedError("3-D oracle: IndRNN layers")
            hprev = hx[li].permute(1, 0, 2, 3).unsqueeze(0)                                      # rnn_cells.py:386-389
            pz = _zero_pad(rnn["k"], rnn["d"])
            h = F.relu(F.conv3d(g.unsqueeze(0), p[pre + "rnn.ih.weight"], p.get(pre + "rnn.ih.bias"), padding=pz, dilation=rnn["d"])
                       + p[pre + "rnn.hh"] * hprev)
            hx[li] = h.squeeze(0)                                                                 # :235-236
            g = hx[li]
        g = _conv3d_nonlinear(g, p["final_layer.0.conv_layer.weight"], p.get("final_layer.0.conv_layer.bias"), final["k"], final["d"],
                              final["nl"])
        g = g.permute(1, 2, 3, 0)                                                                 # :242-243
        for li in range(len(hx)):
            hx[li] = hx[li].permute(1, 0, 2, 3)                                                   # :244-245
        eta = eta + g
        etas.append(eta)
    if not cfg.no_dc:
        raise NotImplementedError("3-D oracle: no_dc cascades")
    return etas, hx


def rim_block_forward(p, cfg, pred, masked_kspace, sense, mask, eta=None, hx=None, sigma=1.0, keep_eta=False):
    """rim_block.py:139-269 (2-D branch; dimensionality 3 -> rim_block_forward_3d).  Returns (list of etas | list of k-spaces, hx)."""
    if getattr(cfg, "dimensionality", 2) == 3:
        return rim_block_forward_3d(p, cfg, pred, masked_kspace, sense, mask, eta, hx, sigma, keep_eta)
    if isinstance(pred, list):                      # :185-186
        pred = pred[-1].detach()
    B = masked_kspace.shape[0]
    if hx is None:                                  # :188-193
        hx = [masked_kspace.new_zeros((B, f, *masked_kspace.shape[2:-1])) for f in cfg.recurrent_filters if f != 0]
    else:
        hx = list(hx)
    if eta is None or eta.ndim < 3:                 # :195-211
        if keep_eta:
            eta = pred
        else:
            img = offt.ifft2(pred, cfg.fft_centered, cfg.fft_normalization, cfg.spatial_dims)
            eta = outils.complex_mul(img, outils.complex_conj(sense)).sum(cfg.coil_dim)
    layers, final = cfg.layer_table()
    etas = []
    for _ in range(cfg.time_steps):                 # :217-249
        g = log_likelihood_gradient(eta, masked_kspace, sense, mask, sigma, cfg.fft_centered,
                                    cfg.fft_normalization, cfg.spatial_dims, cfg.coil_dim).contiguous()
        for li, (conv, rnn) in enumerate(layers):
            pre = f"layers.{li}."
            g = conv_nonlinear(g, p[pre + "convs.conv_layer.weight"], p.get(pre + "convs.conv_layer.bias"),
                               conv["k"], conv["d"], conv["nl"])
            hx[li] = _rnn_apply(p, pre + "rnn.", rnn, g, hx[li])
            g = hx[li]
        g = conv_nonlinear(g, p["final_layer.0.conv_layer.weight"], p.get("final_layer.0.conv_layer.bias"),
                           final["k"], final["d"], final["nl"])
        eta = eta + g.permute(0, 2, 3, 1)
        etas.append(eta)
    if cfg.no_dc:                                   # :253-254
        return etas, hx
    zero = torch.zeros(1, 1, 1, 1, 1).to(masked_kspace)
    soft_dc = torch.where(mask, pred - masked_kspace, zero) * p["dc_weight"]       # :256 (mask must be bool)
    out = [masked_kspace - soft_dc - offt.fft2(outils.complex_mul(e.unsqueeze(cfg.coil_dim), sense),
                                               cfg.fft_centered, cfg.fft_normalization, cfg.spatial_dims)
           for e in etas]                           # :257-267
    return out, hx
