"""Oracle: E2EVN cascade block (reference mridc/collections/reconstruction/models/varnet/vn_block.py).  Test infrastructure."""
import torch

from . import fft as offt
from . import unet as ounet
from . import utils as outils


def sens_expand(x, sens_maps, fft_centered, fft_normalization, spatial_dims):
    """vn_block.py:51-69."""
    return offt.fft2(outils.complex_mul(x, sens_maps), fft_centered, fft_normalization, spatial_dims)


def sens_reduce(x, sens_maps, fft_centered, fft_normalization, spatial_dims, coil_dim):
    """vn_block.py:71-87 (keepdim)."""
    x = offt.ifft2(x, fft_centered, fft_normalization, spatial_dims)
    return outils.complex_mul(x, outils.complex_conj(sens_maps)).sum(dim=coil_dim, keepdim=True)


def soft_dc(pred, ref_kspace, mask, dc_weight):
    """vn_block.py:109-110: where(mask.bool(), pred - ref, 0) * dc_weight."""
    zero = torch.zeros(1, 1, 1, 1, 1).to(pred)
    return torch.where(mask.bool(), pred - ref_kspace, zero) * dc_weight


def varnet_block_forward(p, pred, ref_kspace, sens_maps, mask, num_pools, padding_size, normalize=True,
                         fft_centered=True, fft_normalization="ortho", spatial_dims=None, coil_dim=1, no_dc=False,
                         prefix=""):
    """vn_block.py:89-119.  `p` holds `{prefix}dc_weight` and `{prefix}model.unet.*`."""
    sdc = soft_dc(pred, ref_kspace, mask, p[prefix + "dc_weight"])
    eta = sens_reduce(pred, sens_maps, fft_centered, fft_normalization, spatial_dims, coil_dim)
    eta = ounet.norm_unet_forward(p, eta, num_pools, padding_size, normalize, prefix=prefix + "model.unet.")
    eta = sens_expand(eta, sens_maps, fft_centered, fft_normalization, spatial_dims)
    if not no_dc:
        eta = pred - sdc - eta
    return eta
