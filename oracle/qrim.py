"""Oracle: quantitative RIM (reference mridc/collections/quantitative/models/qrim/{utils,qrim_block}.py, qcirim.py).

Test infrastructure.  2-D mode, `use_reconstruction_module: false` (the model-zoo default, base_qcirim_run.yaml:7).
"""
import torch

from . import fft as offt
from . import rim as orim
from . import utils as outils

SCALING = 1e-3          # qrim/utils.py:34,181
DEFAULT_TES = (3.0, 11.5, 20.0, 28.5)   # qrim/utils.py:61


def megre_signal(R2star, S0, B0, phi, TEs, scaling=SCALING):
    """SignalForwardModel.MEGRESignalModel, qrim/utils.py:71-121.  maps [N,H,W] -> [N,E,H,W,2]."""
    out = []
    for te in TEs:
        ft = torch.exp(-te * scaling * R2star)
        c = torch.cos(B0 * scaling * -te)
        s = torch.sin(B0 * scaling * -te)
        out.append(torch.stack((S0 * ft * c - phi * ft * s, S0 * ft * s + phi * ft * c), -1))
    pred = torch.stack(out, 1)
    pred[pred != pred] = 0.0
    return pred


def analytical_log_likelihood_gradient(R2star, S0, B0, phi, TEs, sens, masked_kspace, mask, fft_centered,
                                       fft_normalization, spatial_dims, coil_dim, scaling=SCALING):
    """qrim/utils.py:166-295 for ONE batch element.  maps [H,W]; sens [C,H,W,2]; masked_kspace [E,C,H,W,2]; mask
    broadcastable to it.  Returns [4,H,W] = (R2*_re, S0_re, R2*_im, S0_im)."""
    R2s, S0_, B0_, phi_ = R2star.unsqueeze(0), S0.unsqueeze(0), B0.unsqueeze(0), phi.unsqueeze(0)
    pred = megre_signal(R2s, S0_, B0_, phi_, TEs, scaling)                                   # [1,E,H,W,2]
    S = sens.unsqueeze(0).unsqueeze(coil_dim - 1)                                             # [1,1,C,H,W,2]
    x = outils.complex_mul(pred.unsqueeze(coil_dim), S)                                       # expand_op :158-163
    x[x != x] = 0
    k = offt.fft2(x, fft_centered, fft_normalization, spatial_dims)
    diff = (k - masked_kspace) * mask                                                         # :242
    dinv = outils.sense(offt.ifft2(diff, fft_centered, fft_normalization, spatial_dims), S, coil_dim)   # [1,E,H,W,2]
    s0d, r2d = [], []
    for te in TEs:                                                                            # :250-275
        ft = torch.exp(-te * scaling * R2s)
        c = torch.cos(B0_ * scaling * -te)
        s = torch.sin(B0_ * scaling * -te)
        s0d.append(torch.stack((ft * c, -ft * s), -1))
        r2d.append(torch.stack((-te * scaling * ft * (S0_ * c - phi_ * s), -te * scaling * ft * (-S0_ * s - phi_ * c)), -1))
    s0d, r2d = torch.stack(s0d, 1), torch.stack(r2d, 1)
    s0_re = dinv[..., 0] * s0d[..., 0] - dinv[..., 1] * s0d[..., 1]
    s0_im = dinv[..., 0] * s0d[..., 1] + dinv[..., 1] * s0d[..., 0]
    r2_re = dinv[..., 0] * r2d[..., 0] - dinv[..., 1] * r2d[..., 1]
    r2_im = dinv[..., 0] * r2d[..., 1] + dinv[..., 1] * r2d[..., 0]
    s0g = torch.stack([s0_re, s0_im], -1).squeeze(0).mean(0)                                  # :290-293 (mean over echoes)
    r2g = torch.stack([r2_re, r2_im], -1).squeeze(0).mean(0)
    return torch.stack([r2g[..., 0], s0g[..., 0], r2g[..., 1], s0g[..., 1]], 0)               # :295


def qrim_block_forward(p, cfg, masked_kspace, R2star_init, S0_init, B0_init, phi_init, TEs, sens, mask, gamma, eta=None, hx=None):
    """qRIMBlock.forward, qrim_block.py:134-240.  `cfg`: orim.RIMConfig with depth-4 input (8 channels).  Returns (etas, None)."""
    B = masked_kspace.shape[0]
    if eta is None:
        eta = torch.stack([R2star_init, S0_init, B0_init, phi_init], dim=1)                   # :188-189
    if hx is None:
        hx = [eta.new_zeros((eta.size(0), f, *eta.size()[2:])) for f in cfg.recurrent_filters if f != 0]
    R2s, S0_, B0_, phi_ = R2star_init * gamma[0], S0_init * gamma[1], B0_init * gamma[2], phi_init * gamma[3]   # :198-201
    layers, final = cfg.layer_table()
    etas = []
    for _ in range(cfg.time_steps):
        grad = torch.zeros_like(eta)
        for i in range(B):                                                                    # :206-224
            grad[i] = analytical_log_likelihood_gradient(R2s[i], S0_[i], B0_[i], phi_[i], TEs, sens[i], masked_kspace[i],
                                                         mask[i], cfg.fft_centered, cfg.fft_normalization, cfg.spatial_dims,
                                                         cfg.coil_dim) / 100
            grad[grad != grad] = 0.0
        g = torch.cat([grad, eta], dim=cfg.coil_dim - 1)                                      # :226
        for li, (conv, rnn) in enumerate(layers):
            pre = f"layers.{li}."
            g = orim.conv_nonlinear(g, p[pre + "convs.conv_layer.weight"], p.get(pre + "convs.conv_layer.bias"), conv["k"],
                                    conv["d"], conv["nl"])
            hx[li] = orim._rnn_apply(p, pre + "rnn.", rnn, g, hx[li])
            g = hx[li]
        g = orim.conv_nonlinear(g, p["final_layer.0.conv_layer.weight"], p.get("final_layer.0.conv_layer.bias"), final["k"],
                                final["d"], final["nl"])
        eta = eta + g
        eta[:, 0] = torch.clamp(eta[:, 0], min=0)                                             # :234-236
        etas.append(eta)
    return etas, None


def rescale_reverse(data, gamma):
    """RescaleByMax.reverse, qrim/utils.py:22-25 (indexes gamma by the BATCH index -- reproduced as is)."""
    return torch.stack([data[i] * gamma[i] for i in range(data.shape[0])], 0)


def qcirim_forward(p, cfg, R2star_init, S0_init, B0_init, phi_init, TEs, y, sens, mask_brain, sampling_mask):
    """qCIRIM.forward with use_reconstruction_module false, qcirim.py:248-312.  Returns
    [pred, cascades_R2star, cascades_S0, cascades_B0, cascades_phi], each list[cascade][time_step] of [B,H,W]."""
    gamma = torch.tensor(cfg["quantitative_module_gamma_regularization_factors"], dtype=torch.float32)
    rcfg = orim.RIMConfig(
        recurrent_layer=cfg["quantitative_module_recurrent_layer"], conv_filters=cfg["quantitative_module_conv_filters"],
        conv_kernels=cfg["quantitative_module_conv_kernels"], conv_dilations=cfg["quantitative_module_conv_dilations"],
        conv_bias=cfg["quantitative_module_conv_bias"], recurrent_filters=cfg["quantitative_module_recurrent_filters"],
        recurrent_kernels=cfg["quantitative_module_recurrent_kernels"],
        recurrent_dilations=cfg["quantitative_module_recurrent_dilations"], recurrent_bias=cfg["quantitative_module_recurrent_bias"],
        depth=cfg["quantitative_module_depth"], time_steps=cfg["quantitative_module_time_steps"], conv_dim=2, no_dc=True,
        fft_centered=cfg["fft_centered"], fft_normalization=cfg["fft_normalization"], spatial_dims=cfg.get("spatial_dims"),
        coil_dim=cfg["coil_dim"])
    r2, s0, b0, ph = R2star_init / gamma[0], S0_init / gamma[1], B0_init / gamma[2], phi_init / gamma[3]     # :248-251
    eta, hx = None, None
    out = [[], [], [], []]
    ncas = cfg["quantitative_module_num_cascades"]
    for i in range(ncas):
        pw = {k[len(f"qcirim.{i}."):]: v for k, v in p.items() if k.startswith(f"qcirim.{i}.")}
        prediction, hx = qrim_block_forward(pw, rcfg, y, r2, s0, b0, ph, TEs, sens, sampling_mask, gamma, eta, hx)
        r2, s0, b0, ph = prediction[-1][:, 0], prediction[-1][:, 1], prediction[-1][:, 2], prediction[-1][:, 3]
        steps = [[], [], [], []]
        for pred in prediction:                                                               # :297-304
            x = rescale_reverse(torch.abs(pred), gamma)
            for m in range(4):
                steps[m].append(x[:, m])
        for m in range(4):
            out[m].append(steps[m])
    return [torch.empty([])] + out
