"""Oracle: SSIM loss and per-slice metrics (reference common/losses/ssim.py, common/metrics/reconstruction_metrics.py,
models/base.py:415-436).  Test infrastructure."""
import numpy as np
import torch
import torch.nn.functional as F


def ssim_loss(X, Y, data_range, win_size=7, k1=0.01, k2=0.03):
    """losses/ssim.py:28-61.  X,Y [B,1,h,w]; data_range [B].  Returns 1 - mean(SSIM map)."""
    w = torch.ones(1, 1, win_size, win_size).to(X) / win_size ** 2
    NP = win_size ** 2
    cov_norm = NP / (NP - 1)
    data_range = data_range[:, None, None, None]
    C1 = (k1 * data_range) ** 2
    C2 = (k2 * data_range) ** 2
    ux, uy = F.conv2d(X, w), F.conv2d(Y, w)
    uxx, uyy, uxy = F.conv2d(X * X, w), F.conv2d(Y * Y, w), F.conv2d(X * Y, w)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    return 1 - ((A1 * A2) / (B1 * B2)).mean()


def mse(gt, pred):
    """reconstruction_metrics.py:11-13."""
    return float(np.mean((gt - pred) ** 2))


def nmse(gt, pred):
    """reconstruction_metrics.py:16-18."""
    return float(np.linalg.norm(gt - pred) ** 2 / np.linalg.norm(gt) ** 2)


def psnr(gt, pred, maxval=None):
    """reconstruction_metrics.py:21-25 (skimage peak_signal_noise_ratio = 10 log10(range^2 / mse))."""
    maxval = np.max(gt) if maxval is None else maxval
    return float(10 * np.log10((maxval ** 2) / np.mean((gt.astype(np.float64) - pred.astype(np.float64)) ** 2)))


def ssim(gt, pred, maxval=None):
    """reconstruction_metrics.py:28-41.  skimage's structural_similarity (7x7 uniform window, sample
    covariance, border crop) equals 1 - ssim_loss on the valid region; restated through ssim_loss."""
    if gt.ndim != 3:
        raise ValueError("Unexpected number of dimensions in ground truth.")
    if gt.ndim != pred.ndim:
        raise ValueError("Ground truth dimensions does not match pred.")
    maxval = np.max(gt) if maxval is None else maxval
    tot = 0.0
    for s in range(gt.shape[0]):
        X = torch.from_numpy(np.ascontiguousarray(gt[s])).double()[None, None]
        Y = torch.from_numpy(np.ascontiguousarray(pred[s])).double()[None, None]
        tot += 1.0 - float(ssim_loss(X, Y, torch.tensor([float(maxval)], dtype=torch.float64)))
    return tot / gt.shape[0]


def postprocess(pred_complex, target):
    """models/base.py:415-419: abs, divide by max -- what test_step feeds the metrics."""
    out = torch.abs(pred_complex).detach().cpu()
    out = out / out.max()
    tgt = torch.abs(target).detach().cpu()
    tgt = tgt / tgt.max()
    return out, tgt


def slice_metrics(pred_complex, target):
    """models/base.py:427-436: metrics with maxval = output.max() - output.min()."""
    out, tgt = postprocess(pred_complex, target)
    o, t = out.numpy(), tgt.numpy()
    mv = o.max() - o.min()
    return dict(mse=mse(t, o), nmse=nmse(t, o), ssim=ssim(t, o, maxval=mv), psnr=psnr(t, o, maxval=mv))
