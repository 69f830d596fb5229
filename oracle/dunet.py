"""Oracle: DUNet = SensitivityNetwork(DIDN regulariser under a complex instance normalisation, sigmanet data-consistency layer)
(reference models/dunet.py, models/didn/didn.py, models/sigmanet/sensitivity_net.py).  Test infrastructure: plain torch on the CPU, a
functional restatement over a state_dict; pinned by tests/golden/g21_dunet.npz (generated from the imported reference blocks).

Shapes as the reference runs them (its own test, tests/collections/reconstruction/models/test_dunet.py, uses batch 1): the network is
handed the coil-combined image [1,H,W,2]; the normalisation wrapper broadcasts its [B,1,1,1] statistics against [B,H,W] planes, which
for B = 1 yields [1,1,H,W,2], and everything downstream (x - R(x), the data layers) carries that 5-D shape.  The covariance divisor is
the reference's `shape[2] * shape[3] - 1` of the 4-D input, i.e. 2 W - 1 -- restated as is."""
import torch
import torch.nn.functional as F

from . import dc_layers as odc
from . import fft as offt
from . import utils as outils


# ---- DIDN (didn/didn.py) ------------------------------------------------------------------------------------------------------------
def _conv(p, pre, x, stride=1, padding=1):
    return F.conv2d(x, p[pre + "weight"], p[pre + "bias"], stride=stride, padding=padding)


def _conv_prelu(p, pre, x):
    """nn.Sequential(conv, PReLU) registered as `<pre>0` / `<pre>1` (didn.py:62-69)."""
    return F.prelu(_conv(p, pre + "0.", x), p[pre + "1.weight"])


def _subpixel(p, pre, x):
    """didn.py:23-39: 1x1 conv into 4x the channels, PixelShuffle(2)."""
    return F.pixel_shuffle(_conv(p, pre + "conv.", x, padding=0), 2)


def _crop(x, shape):
    """didn.py:164-183."""
    return x[:, :, : min(x.shape[-2], shape[0]), : min(x.shape[-1], shape[1])]


def _dub_pad(x):
    """didn.py:143-162: reflect-pad odd heights / widths by one."""
    padding = [0, 0, 0, 0]
    if x.shape[-2] % 2 != 0:
        padding[3] = 1
    if x.shape[-1] % 2 != 0:
        padding[1] = 1
    return F.pad(x, padding, "reflect") if sum(padding) else x


def dub_forward(p, pre, x):
    """didn.py:184-208.  conv1_1 / conv1_2 are `Sequential(*[conv, PReLU] * 2)`: the SAME two modules applied twice (didn.py:115, 137)."""
    x1 = _dub_pad(x)
    x1 = x1 + _conv_prelu(p, pre + "conv1_1.", _conv_prelu(p, pre + "conv1_1.", x1))
    x2 = _conv(p, pre + "down1.", x1, stride=2)
    x2 = x2 + _conv_prelu(p, pre + "conv2_1.", x2)
    out = _conv(p, pre + "down2.", x2, stride=2)
    out = out + _conv_prelu(p, pre + "conv3_1.", out)
    out = _subpixel(p, pre + "up1.0.", out)
    out = torch.cat([x2, _crop(out, x2.shape[-2:])], 1)
    out = _conv(p, pre + "conv_agg_1.", out, padding=0)
    out = out + _conv_prelu(p, pre + "conv2_2.", out)
    out = _subpixel(p, pre + "up2.0.", out)
    out = torch.cat([x1, _crop(out, x1.shape[-2:])], 1)
    out = _conv(p, pre + "conv_agg_2.", out, padding=0)
    out = out + _conv_prelu(p, pre + "conv1_2.", _conv_prelu(p, pre + "conv1_2.", out))
    return x + _crop(_conv_prelu(p, pre + "conv_out.", out), x.shape[-2:])


def recon_block_forward(p, pre, x, num_convs):
    """didn.py:73-86."""
    out = x
    for i in range(num_convs - 1):
        out = _conv_prelu(p, f"{pre}convs.{i}.", out)
    return x + _conv(p, f"{pre}convs.{num_convs - 1}.", out)


def didn_forward(p, x, num_dubs, num_convs_recon, prefix="", skip_connection=False):
    """didn.py:300-325."""
    out = _conv_prelu(p, prefix + "conv_in.", x)
    out = _conv(p, prefix + "down.", out, stride=2)
    dub_outs = []
    for i in range(num_dubs):
        out = dub_forward(p, f"{prefix}dubs.{i}.", out)
        dub_outs.append(out)
    out = torch.cat([recon_block_forward(p, prefix + "recon_block.", d, num_convs_recon) for d in dub_outs], 1)
    out = _conv(p, prefix + "recon_agg.", out, padding=0)
    out = _conv_prelu(p, prefix + "conv.", out)
    out = _subpixel(p, prefix + "up2.", out)
    out = _crop(_conv(p, prefix + "conv_out.", out), x.shape[-2:])
    return x + out if skip_connection else out


# ---- complex instance normalisation around the regulariser (sensitivity_net.py:17-139) --------------------------------------------------
def complex_pseudocovariance_half(data):
    """sensitivity_net.py:38-78 on mean-free data of ANY rank >= 3 ending in 2: N = shape[2] * shape[3], sums over dims 1 .. rank-2."""
    shape = data.shape
    N = shape[2] * shape[3]
    re, im = torch.unbind(data, dim=-1)
    dim = list(range(1, len(shape) - 1))
    cxx = (re * re).sum(dim=dim, keepdim=True) / (N - 1)
    cyy = (im * im).sum(dim=dim, keepdim=True) / (N - 1)
    cxy = (re * im).sum(dim=dim, keepdim=True) / (N - 1)
    root = torch.sqrt((cxx + cyy) ** 2 / 4 - cxx * cyy + cxy ** 2)
    s1, s2 = (cxx + cyy) / 2 - root, (cxx + cyy) / 2 + root
    v1x, v1y, v2x, v2y = s1 - cyy, cxy, s2 - cyy, cxy
    norm1 = torch.sqrt(torch.sum(v1x * v1x + v1y * v1y, dim=dim, keepdim=True))
    norm2 = torch.sqrt(torch.sum(v2x * v2x + v2y * v2y, dim=dim, keepdim=True))
    v1x, v1y, v2x, v2y = v1x / norm1, v1y / norm1, v2x / norm2, v2y / norm2
    det = v1x * v2y - v2x * v1y
    s1, s2 = torch.sqrt(s1) / det, torch.sqrt(s2) / det
    return (v1x * v2y * s1 - v1y * v2x * s2, v1x * v2x * (s2 - s1), v1y * v2y * (s1 - s2), v1x * v2y * s2 - v1y * v2x * s1)   # xx, xy, yx, yy


def complex_norm_wrapper(model_fn, x):
    """ComplexNormWrapper.forward (sensitivity_net.py:128-139) with ComplexInstanceNorm.set_normalization / normalize / unnormalize
    (:85-118).  x [B,H,W,2] -> [B,B,H,W,2] by the reference's broadcasting (B = 1: [1,1,H,W,2]); a 5-D x keeps its shape."""
    mean = torch.mean(x).reshape(1, 1, 1, 1)
    xx, xy, yx, yy = (c.reshape(-1, 1, 1, 1) for c in complex_pseudocovariance_half(x - torch.mean(x)))
    re, im = torch.unbind(x - mean, dim=-1)
    det = xx * yy - xy * yx
    ixx, ixy, iyx, iyy = yy / det, -xy / det, -yx / det, xx / det
    out = torch.stack([ixx * re + ixy * im, iyx * re + iyy * im], -1).clamp(-6, 6)
    shp = out.shape
    out = model_fn(out.reshape(shp[0] * shp[1], *shp[2:]).permute(0, 3, 1, 2))
    out = out.permute(0, 2, 3, 1).reshape(*shp)
    re, im = torch.unbind(out, dim=-1)
    return torch.stack([xx * re + xy * im, yx * re + yy * im], -1) + mean


def sensitivity_network_forward(p, cfg, x, y, smaps, mask, prefix="model."):
    """SensitivityNetwork.forward (sensitivity_net.py:186-212) with a DIDN regulariser and the configured data layer."""
    n_total, shared = cfg["num_iter"], cfg["shared_params"]
    n_mod = 1 if shared else n_total
    c, n, sd = cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims")
    term = cfg["data_consistency_term"]
    for i in range(n_total):
        j = i % n_mod
        reg = complex_norm_wrapper(lambda t: didn_forward(p, t, cfg["didn_num_dubs"], cfg["didn_num_convs_recon"],
                                                           prefix=f"{prefix}gradR.{j}.model."), x)
        x_thalf = x - reg
        if term == "GD":
            x = odc.data_gd(x_thalf, y, smaps, mask, p[f"{prefix}gradD.{j}.data_weight"], c, n, sd)
        elif term == "PROX":
            x = odc.data_prox_cg(x_thalf, y, smaps, mask, p[f"{prefix}gradD.{j}.lambdaa"], 1e-6, cfg["data_consistency_iterations"], c, n, sd)
        elif term == "VS":
            x = odc.data_vs(x_thalf, y, smaps, mask, p[f"{prefix}gradD.{j}.alpha"], p[f"{prefix}gradD.{j}.beta"], c, n, sd)
        else:
            x = x_thalf
    return x


def dunet_forward(p, cfg, y, sensitivity_maps, mask, init_pred, target):
    """dunet.py:162-176."""
    c, n, sd, cd = cfg["fft_centered"], cfg["fft_normalization"], cfg.get("spatial_dims"), cfg.get("coil_dim", 1)
    init_pred = outils.complex_mul(offt.ifft2(y, c, n, sd), outils.complex_conj(sensitivity_maps)).sum(cd)
    image = sensitivity_network_forward(p, cfg, init_pred, y, sensitivity_maps, mask)
    image = outils.complex_mul(image, outils.complex_conj(sensitivity_maps)).sum(cd)
    image = torch.view_as_complex(image.contiguous())
    _, image = outils.center_crop_to_smallest(target, image)
    return image
