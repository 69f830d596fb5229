"""Seeded synthetic fastMRI-shaped multicoil data for bench.py / smoke / large-size tests.

Image: sum of random ellipses x smooth phase; sensitivity maps: Gaussian coil profiles on a circle with a linear
phase, normalised so sum_c |S_c|^2 = 1; k-space = fft2(img * S) (non-centred, backward) + complex noise; mask: 1-D random
columns (center_fraction 0.08, acceleration 4, same recipe as the reference's RandomMaskFunc, subsample.py:137-153);
everything max-normalised like reconstruction/parts/transforms.py:572-617.  Generated on the host with numpy.
"""
import numpy as np
import torch


def random_mask_1d(num_cols, center_fraction=0.08, acceleration=4, seed=123):
    """RandomMaskFunc recipe (reference subsample.py:137-153) -> bool [num_cols]."""
    rng = np.random.RandomState()
    rng.seed(seed)
    rng.randint(0, 1)                                   # choose_acceleration with one choice
    num_low = int(round(num_cols * center_fraction))
    prob = (num_cols / acceleration - num_low) / (num_cols - num_low)
    mask = rng.uniform(size=num_cols) < prob
    pad = (num_cols - num_low + 1) // 2
    mask[pad:pad + num_low] = True
    return mask


def make_slice(C=15, H=640, W=372, slice_idx=0, noise=1e-3, mask_dtype=torch.bool):
    """Returns dict(y [1,C,H,W,2], sensitivity_maps [1,C,H,W,2], mask [1,1,1,W,1], target [1,H,W], kspace)."""
    rng = np.random.default_rng(1234 + slice_idx)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.zeros((H, W))
    for _ in range(6):
        cx, cy = rng.uniform(-0.4, 0.4, 2)
        ax, ay = rng.uniform(0.15, 0.6, 2)
        th = rng.uniform(0, np.pi)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img += rng.uniform(0.2, 1.0) * ((xr / ax) ** 2 + (yr / ay) ** 2 <= 1.0)
    phase = np.exp(1j * (0.8 * xx + 0.5 * yy * yy + rng.uniform(0, 1)))
    img = img * phase
    ang = 2 * np.pi * np.arange(C) / C
    S = np.stack([np.exp(-((xx - 0.8 * np.cos(a)) ** 2 + (yy - 0.8 * np.sin(a)) ** 2) / (2 * 0.7 ** 2))
                  * np.exp(1j * (1.5 * (xx * np.cos(a) + yy * np.sin(a)))) for a in ang])
    S = S / np.sqrt((np.abs(S) ** 2).sum(0, keepdims=True))
    k = np.fft.fft2(img[None] * S, axes=(-2, -1))
    k = k + noise * np.abs(k).max() * (rng.standard_normal(k.shape) + 1j * rng.standard_normal(k.shape))
    # max-normalise in image space (transforms.py:528-543): k /= max |ifft2(k)|
    k = k / np.abs(np.fft.ifft2(k, axes=(-2, -1))).max()
    m = random_mask_1d(W)
    y = k * m[None, None, :]

    def tt(a):
        return torch.from_numpy(np.stack([a.real, a.imag], -1).astype(np.float32))

    S_t = tt(S)[None]
    S_t = S_t / torch.sqrt((S_t ** 2).sum(-1)).max()   # transforms.py:610-612
    target = np.abs((np.fft.ifft2(k, axes=(-2, -1)) * np.conj(S)).sum(0))
    target = target / target.max()
    mask = torch.from_numpy(m).reshape(1, 1, 1, W, 1)
    mask = mask if mask_dtype == torch.bool else mask.to(mask_dtype)
    return dict(y=tt(y)[None], kspace=tt(k)[None], sensitivity_maps=S_t, mask=mask,
                target=torch.from_numpy(target.astype(np.float32))[None])


CIRIM_BASELINE_CFG = dict(
    # projects/reconstruction/model_zoo/conf/base_cirim_run.yaml:5-63 with BASELINE.json's 8 cascades x 5 (->8) time-steps
    recurrent_layer="IndRNN", conv_filters=[64, 64, 2], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1],
    conv_bias=[True, True, False], recurrent_filters=[64, 64, 0], recurrent_kernels=[1, 1, 0],
    recurrent_dilations=[1, 1, 0], recurrent_bias=[True, True, False], depth=2, time_steps=5, conv_dim=2,
    num_cascades=8, no_dc=True, keep_eta=True, accumulate_estimates=True, use_sens_net=False,
    coil_combination_method="SENSE", fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1],
    coil_dim=1, dimensionality=2, train_loss_fn="l1", val_loss_fn="l1")

E2EVN_BASELINE_CFG = dict(
    # base_vn_run.yaml:5-29 with BASELINE.json's 6 cascades
    num_cascades=6, channels=14, pooling_layers=2, padding_size=11, normalize=True, no_dc=False, use_sens_net=False,
    coil_combination_method="SENSE", fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1,
    dimensionality=2, train_loss_fn="l1", val_loss_fn="l1")
