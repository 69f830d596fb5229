"""Oracle: CascadeNet (reference models/conv/conv2d.py, models/cascadenet/ccnn_block.py, models/ccnn.py).  Test infrastructure."""
import torch
import torch.nn.functional as F

from . import fft as offt
from . import utils as outils
from . import varnet as ovn


def conv2d_stack_forward(p, x, n_convs, batchnorm=False, prefix="conv."):
    """conv/conv2d.py:34-69: [Conv2d(3x3, padding 1) (+ BatchNorm2d eps 1e-4) (+ PReLU, not after the last)] * n_convs.
    `p` uses the nn.Sequential indices of the reference (`conv.{i}.weight` ...)."""
    if x.dim() == 5:                                     # conv2d.py:64-67
        x = x.squeeze(1)
        if x.shape[-1] == 2:
            x = x.permute(0, 3, 1, 2)
    idx = 0
    for i in range(n_convs):
        x = F.conv2d(x, p[f"{prefix}{idx}.weight"], p[f"{prefix}{idx}.bias"], padding=1)
        idx += 1
        if batchnorm:
            x = F.batch_norm(x, p[f"{prefix}{idx}.running_mean"], p[f"{prefix}{idx}.running_var"], p[f"{prefix}{idx}.weight"],
                             p[f"{prefix}{idx}.bias"], training=False, eps=1e-4)
            idx += 1
        if i != n_convs - 1:
            x = F.prelu(x, p[f"{prefix}{idx}.weight"])
            idx += 1
    return x


def cascadenet_block_forward(p, pred, ref_kspace, sens_maps, mask, n_convs, batchnorm=False, fft_centered=True,
                             fft_normalization="ortho", spatial_dims=None, coil_dim=1, no_dc=False, prefix=""):
    """ccnn_block.py:101-139.  `p` holds `{prefix}dc_weight` and `{prefix}model.conv.*`."""
    sdc = ovn.soft_dc(pred, ref_kspace, mask, p[prefix + "dc_weight"])
    eta = ovn.sens_reduce(pred, sens_maps, fft_centered, fft_normalization, spatial_dims, coil_dim)
    eta = conv2d_stack_forward(p, eta.squeeze(coil_dim).permute(0, 3, 1, 2), n_convs, batchnorm,
                               prefix=prefix + "model.conv.").permute(0, 2, 3, 1)
    if eta.dim() < sens_maps.dim():
        eta = eta.unsqueeze(1)
    eta = ovn.sens_expand(eta, sens_maps, fft_centered, fft_normalization, spatial_dims)
    if not no_dc:
        eta = pred - sdc - eta
    return eta


def cascadenet_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """ccnn.py:116-142."""
    pred = y.clone()
    for i in range(cfg["num_cascades"]):
        pred = cascadenet_block_forward(p, pred, y, sensitivity_maps, mask, cfg["n_convs"], cfg.get("batchnorm", False),
                                        cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"),
                                        cfg.get("coil_dim", 1), cfg["no_dc"], prefix=f"cascades.{i}.")
    pred = offt.ifft2(pred, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"))
    pred = outils.coil_combination(pred, sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE"),
                                   dim=cfg.get("coil_dim", 1))
    pred = torch.view_as_complex(pred.contiguous())
    _, pred = outils.center_crop_to_smallest(target, pred)
    return pred
