"""Oracle: Recurrent Variational Network (reference models/recurrentvarnet/conv2gru.py, recurrentvarnet.py, models/rvn.py).
Test infrastructure."""
import math

import torch
import torch.nn.functional as F

from . import fft as offt
from . import utils as outils


def _rep_conv(x, w, b, dilation):
    """ReplicationPad2d(dilation * (k - 1) / 2) + Conv2d(padding 0, dilation) (conv2gru.py:64-80, recurrentvarnet.py:68-71)."""
    pad = dilation * (w.shape[-1] - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="replicate")
    return F.conv2d(x, w, b, dilation=dilation)


def conv2dgru_forward(p, cell_input, previous_state, num_layers, hidden_channels, prefix=""):
    """conv2gru.py:112-163 with dense_connect = 0, replication padding, no instance norm (the RecurrentVarNetBlock's settings).
    Layer idx: conv 5x5 (idx 0) / 3x3 dilation 2 (idx 1) / 3x3 (others) + ReLU, then the GRU on 1x1 (or k x k, zero padded)
    convolutions of cat(input, state): update, reset, candidate on cat(input, state * reset)."""
    if previous_state is None:
        B, _, H, W = cell_input.shape
        previous_state = torch.zeros(B, hidden_channels, H, W, num_layers, dtype=cell_input.dtype)
    new_states = []
    for idx in range(num_layers):
        w, b = p[f"{prefix}conv_blocks.{idx}.1.weight"], p[f"{prefix}conv_blocks.{idx}.1.bias"]
        cell_input = F.relu(_rep_conv(cell_input, w, b, 2 if idx == 1 else 1))
        h = previous_state[..., idx]
        stacked = torch.cat([cell_input, h], dim=1)

        def gate(name, inp):
            gw, gb = p[f"{prefix}{name}.{idx}.0.weight"], p[f"{prefix}{name}.{idx}.0.bias"]
            return F.conv2d(inp, gw, gb, padding=gw.shape[-1] // 2)

        update = torch.sigmoid(gate("update_gates", stacked))
        reset = torch.sigmoid(gate("reset_gates", stacked))
        delta = torch.tanh(gate("out_gates", torch.cat([cell_input, h * reset], dim=1)))
        cell_input = h * (1 - update) + delta * update
        new_states.append(cell_input)
        cell_input = F.relu(cell_input)
    w, b = p[f"{prefix}conv_blocks.{num_layers}.1.weight"], p[f"{prefix}conv_blocks.{num_layers}.1.bias"]
    out = _rep_conv(cell_input, w, b, 2 if num_layers == 1 else 1)
    return out, torch.stack(new_states, dim=-1)


def recurrent_init_forward(p, x, dilations, depth, multiscale_depth=1, prefix=""):
    """recurrentvarnet.py:17-108."""
    feats = []
    for i, dil in enumerate(dilations):
        x = F.relu(_rep_conv(x, p[f"{prefix}conv_blocks.{i}.1.weight"], p[f"{prefix}conv_blocks.{i}.1.bias"], dil))
        if multiscale_depth > 1:
            feats.append(x)
    if multiscale_depth > 1:
        x = torch.cat(feats[-multiscale_depth:], dim=1)
    outs = [F.relu(F.conv2d(x, p[f"{prefix}out_blocks.{j}.0.weight"], p[f"{prefix}out_blocks.{j}.0.bias"])) for j in range(depth)]
    return torch.stack(outs, dim=-1)


def rvn_block_forward(p, current_kspace, masked_kspace, sampling_mask, sensitivity_map, hidden_state, num_layers, hidden_channels,
                      fft_centered=True, fft_normalization="ortho", spatial_dims=None, coil_dim=1, prefix=""):
    """recurrentvarnet.py:163-240."""
    kspace_error = torch.where(sampling_mask == 0, torch.tensor([0.0], dtype=masked_kspace.dtype), current_kspace - masked_kspace)
    term = torch.cat([outils.complex_mul(offt.ifft2(k, fft_centered, fft_normalization, spatial_dims),
                                         outils.complex_conj(sensitivity_map)).sum(coil_dim)
                      for k in torch.split(current_kspace, 2, -1)], dim=-1).permute(0, 3, 1, 2)
    term, hidden_state = conv2dgru_forward(p, term, hidden_state, num_layers, hidden_channels, prefix=prefix + "regularizer.")
    term = term.permute(0, 2, 3, 1)
    term = torch.cat([offt.fft2(outils.complex_mul(img.unsqueeze(coil_dim), sensitivity_map), fft_centered, fft_normalization,
                                spatial_dims) for img in torch.split(term, 2, -1)], dim=-1)
    return current_kspace - p[prefix + "learning_rate"] * kspace_error + term, hidden_state


def rvn_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """rvn.py:163-226 (learned initializer with `sense` initialisation, or none)."""
    c, n, sd, cd = cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"), cfg.get("coil_dim", 1)
    steps = 8 * math.ceil(cfg["num_steps"] / 8)               # rvn.py:51
    state = None
    if cfg.get("learned_initializer"):
        if cfg["initializer_initialization"] != "sense":
            raise NotImplementedError("oracle restates the `sense` initialisation")
        img = outils.complex_mul(offt.ifft2(y, c, n, sd), outils.complex_conj(sensitivity_maps)).sum(cd).unsqueeze(cd)
        state = recurrent_init_forward(p, offt.fft2(img, c, n, sd).sum(1).permute(0, 3, 1, 2), cfg["initializer_dilations"],
                                       cfg["recurrent_num_layers"], cfg.get("initializer_multiscale", 1), prefix="initializer.")
    k = y.clone()
    for step in range(steps):
        bi = step if cfg["no_parameter_sharing"] else 0
        k, state = rvn_block_forward(p, k, y, mask, sensitivity_maps, state, cfg["recurrent_num_layers"],
                                     cfg["recurrent_hidden_channels"], c, n, sd, cd, prefix=f"block_list.{bi}.")
    eta = offt.ifft2(k, c, n, sd)
    eta = torch.view_as_complex(outils.coil_combination(eta, sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE"),
                                                        dim=cd).contiguous())
    _, eta = outils.center_crop_to_smallest(target, eta)
    return eta
