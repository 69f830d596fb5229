"""Oracle: VSNet (reference models/variablesplittingnet/vsnet_block.py, models/vsnet.py).  Test infrastructure."""
import torch

from . import cascadenet as occ
from . import fft as offt
from . import utils as outils


def data_consistency(pred_kspace, ref_kspace, mask, dc_weight):
    """vsnet_block.py:23-25: hard replacement of the sampled locations, times dc_weight."""
    return ((1 - mask) * pred_kspace + mask * ref_kspace) * dc_weight


def weighted_average(x, Sx, param):
    """vsnet_block.py:35-36."""
    return param * x + (1 - param) * Sx


def vsnet_block_forward(p, kspace, sens_maps, mask, num_cascades, n_convs, fft_centered=True, fft_normalization="ortho",
                        spatial_dims=None, coil_dim=1, prefix=""):
    """vsnet_block.py:118-146.  The denoiser / DC / averaging modules are ONE instance each, listed num_cascades times
    (vsnet.py:81-83), so the state_dict repeats the same values under `denoiser_block.{i}.`; index 0 is read here."""
    def sens_reduce(x):
        x = offt.ifft2(x, fft_centered, fft_normalization, spatial_dims)
        return outils.complex_mul(x, outils.complex_conj(sens_maps)).sum(coil_dim)

    for idx in range(num_cascades):
        pred = sens_reduce(kspace)
        pred = occ.conv2d_stack_forward(p, pred.permute(0, 3, 1, 2), n_convs, False,
                                        prefix=f"{prefix}denoiser_block.{idx}.conv.").permute(0, 2, 3, 1)
        pred = offt.fft2(outils.complex_mul(pred, sens_maps), fft_centered, fft_normalization, spatial_dims)
        sx = data_consistency(pred, kspace, mask, p[f"{prefix}data_consistency_block.{idx}.dc_weight"])
        sx = sens_reduce(sx)
        kspace = weighted_average(kspace + pred, sx, p[f"{prefix}weighted_average_block.{idx}.param"])
    return kspace


def vsnet_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """vsnet.py:143-167 (CONV denoiser)."""
    image = vsnet_block_forward(p, y, sensitivity_maps, mask, cfg["num_cascades"], cfg["imspace_conv_n_convs"],
                                cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"), cfg.get("coil_dim", 1),
                                prefix="model.")
    image = outils.coil_combination(offt.ifft2(image, cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims")),
                                    sensitivity_maps, method=cfg.get("coil_combination_method", "SENSE"), dim=cfg.get("coil_dim", 1))
    image = torch.view_as_complex(image.contiguous())
    _, image = outils.center_crop_to_smallest(target, image)
    return image
