"""Oracle: model-level compositions (reference models/cirim.py, vn.py, unet.py, zf.py).  Test infrastructure.

The reference model classes subclass a pytorch-lightning base and cannot be imported where the
goldens are generated; their `forward` bodies are ~20 lines each and are restated here over the
block oracles.  Weight dicts use the reference state_dict keys (`cirim.{i}.*`, `cascades.{i}.*`, `unet.*`).
"""
import math

import torch

from . import fft as offt
from . import rim as orim
from . import unet as ounet
from . import utils as outils
from . import varnet as ovn


def _sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def cirim_time_steps(time_steps):
    """cirim.py:50-51: rounded up to a multiple of 8."""
    return 8 * math.ceil(time_steps / 8)


def process_intermediate_pred(pred, sensitivity_maps, target, no_dc, fft_centered, fft_normalization, spatial_dims,
                              coil_dim, coil_combination_method="SENSE", do_coil_combination=False):
    """cirim.py:167-197."""
    if not no_dc or do_coil_combination:
        pred = offt.ifft2(pred, fft_centered, fft_normalization, spatial_dims)
        pred = outils.coil_combination(pred, sensitivity_maps, method=coil_combination_method, dim=coil_dim)
    pred = torch.view_as_complex(pred.contiguous())
    _, pred = outils.center_crop_to_smallest(target, pred)
    return pred


def cirim_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target, cascade_stamps=None):
    """cirim.py:146-165.  `cfg` is a dict with the reference's YAML keys.  Returns list[cascade][time_step].
    `cascade_stamps`: optional list that receives a time.perf_counter() stamp after every cascade (bench.py's cpu_baseline leg)."""
    rcfg = orim.RIMConfig(
        recurrent_layer=cfg["recurrent_layer"], conv_filters=cfg["conv_filters"], conv_kernels=cfg["conv_kernels"],
        conv_dilations=cfg["conv_dilations"], conv_bias=cfg["conv_bias"], recurrent_filters=cfg["recurrent_filters"],
        recurrent_kernels=cfg["recurrent_kernels"], recurrent_dilations=cfg["recurrent_dilations"],
        recurrent_bias=cfg["recurrent_bias"], depth=cfg.get("depth", 2), time_steps=cirim_time_steps(cfg["time_steps"]),
        conv_dim=cfg.get("conv_dim", 2), no_dc=cfg["no_dc"], fft_centered=cfg["fft_centered"],
        fft_normalization=cfg["fft_normalization"], spatial_dims=cfg.get("spatial_dims"), coil_dim=cfg.get("coil_dim", 1),
        dimensionality=cfg.get("dimensionality", 2))
    prediction = y.clone()
    init_pred = None if init_pred is None or init_pred.dim() < 4 else init_pred
    out = []
    for i in range(cfg["num_cascades"]):
        prediction, _ = orim.rim_block_forward(
            _sub(p, f"cirim.{i}."), rcfg, prediction, y, sensitivity_maps, mask, init_pred, None, 1.0,
            keep_eta=False if i == 0 else cfg["keep_eta"])
        out.append([process_intermediate_pred(e, sensitivity_maps, target, cfg["no_dc"], rcfg.fft_centered,
                                              rcfg.fft_normalization, rcfg.spatial_dims, rcfg.coil_dim,
                                              cfg.get("coil_combination_method", "SENSE")) for e in prediction])
        if cascade_stamps is not None:
            import time
            cascade_stamps.append(time.perf_counter())
    return out


def cirim_process_loss(target, pred, loss_fn, time_steps, num_cascades):
    """cirim.py:199-247 with accumulate_estimates=True and a non-SSIM loss (l1 / mse).

    Reproduces the weighting quirk: every scalar loss is multiplied by the whole logspace vector.
    """
    target = torch.abs(target / torch.max(torch.abs(target)))
    cascades_loss = []
    for cascade_pred in pred:
        ls = [loss_fn(target, torch.abs(t / torch.max(torch.abs(t)))) for t in cascade_pred]
        w = torch.logspace(-1, 0, steps=time_steps).to(ls[0])
        cascades_loss.append(sum(sum([x * w for x in ls]) / time_steps))
    return sum(cascades_loss) / num_cascades


def varnet_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """vn.py:125-142."""
    est = y.clone()
    for i in range(cfg["num_cascades"]):
        est = ovn.varnet_block_forward(
            p, est, y, sensitivity_maps, mask, cfg["pooling_layers"], cfg["padding_size"], cfg.get("normalize", True),
            cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"), cfg.get("coil_dim", 1),
            cfg["no_dc"], prefix=f"cascades.{i}.")
    est = offt.ifft2(est, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"))
    est = outils.coil_combination(est, sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE"),
                                  dim=cfg.get("coil_dim", 1))
    est = torch.view_as_complex(est.contiguous())
    _, est = outils.center_crop_to_smallest(target, est)
    return est


def unet_model_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """unet.py:108-121."""
    cd = cfg.get("coil_dim", 1)
    eta = torch.view_as_complex(outils.coil_combination(
        offt.ifft2(y, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims")),
        sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE"), dim=cd).contiguous())
    _, eta = outils.center_crop_to_smallest(target, eta)
    out = ounet.norm_unet_forward(p, torch.view_as_real(eta.unsqueeze(cd)), cfg["pooling_layers"],
                                  cfg["padding_size"], cfg.get("normalize", True), prefix="unet.unet.")
    return torch.view_as_complex(out).squeeze(cd)


def zf_forward(cfg, y, sensitivity_maps, mask, target=None):
    """zf.py:90-100."""
    pred = outils.coil_combination(
        offt.ifft2(y, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims")),
        sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE").upper(), dim=cfg.get("coil_dim", 1))
    pred = outils.check_stacked_complex(pred.contiguous())
    _, pred = outils.center_crop_to_smallest(target, pred)
    return pred


def sens_net_pad_and_low_freqs(mask, num_low_frequencies=None):
    """base.py:842-884: (pad, num_low_frequencies) per batch element from the mask's contiguous centre block."""
    if num_low_frequencies is None or num_low_frequencies == 0:
        squeezed = mask[:, 0, 0, :, 0].to(torch.int8)
        cent = squeezed.shape[1] // 2
        left = torch.argmin(squeezed[:, :cent].flip(1), dim=1)     # first zero walking outwards = length of the run of ones
        right = torch.argmin(squeezed[:, cent:], dim=1)
        nlf = torch.max(2 * torch.min(left, right), torch.ones_like(left))
    else:
        nlf = num_low_frequencies * torch.ones(mask.shape[0], dtype=mask.dtype, device=mask.device)
    pad = torch.div(mask.shape[-2] - nlf + 1, 2, rounding_mode="trunc")
    return pad, nlf


def batched_mask_center(x, mask_from, mask_to, mask_type="2D"):
    """utils.py:346-410."""
    out = torch.zeros_like(x)
    if mask_from.shape[0] == 1:
        a, b = int(mask_from), int(mask_to)
        if mask_type == "1D":
            out[:, :, :, a:b] = x[:, :, :, a:b]
        else:
            out[:, :, a:b] = x[:, :, a:b]
        return out
    for i, (a, b) in enumerate(zip(mask_from, mask_to)):
        out[i, :, :, a:b] = x[i, :, :, a:b]
    return out


def sens_net_forward(p, cfg, masked_kspace, mask, num_low_frequencies=None):
    """BaseSensitivityModel.forward, base.py:886-932.  `p` holds the NormUnet weights under `norm_unet.unet.`;
    cfg keys: sens_pools, sens_mask_type, sens_normalize, sens_mask_center, padding_size (15), fft_*, spatial_dims, coil_dim."""
    cd = cfg.get("coil_dim", 1)
    if cfg.get("sens_mask_center", True):
        pad, nlf = sens_net_pad_and_low_freqs(mask, num_low_frequencies)
        masked_kspace = batched_mask_center(masked_kspace, pad, pad + nlf, cfg.get("sens_mask_type", "2D"))
    img = offt.ifft2(masked_kspace, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"))
    b, c, h, w, two = img.shape
    out = ounet.norm_unet_forward(p, img.reshape(b * c, 1, h, w, two), cfg["sens_pools"], cfg.get("sens_padding_size", 15),
                                  cfg.get("sens_normalize", True), prefix="norm_unet.unet.")
    out = out.reshape(b, c, h, w, two)
    if cfg.get("sens_normalize", True):
        out = out / outils.rss_complex(out, dim=cd).unsqueeze(-1).unsqueeze(cd)      # base.py:824-840
    return out
