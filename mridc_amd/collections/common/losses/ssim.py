"""SSIM loss -- drop-in for `mridc.collections.common.losses.ssim.SSIMLoss` (reference ssim.py:11-61).

Used for reporting ("SSIM vs ref") on small [B,1,h,w] magnitude images after the reconstruction; the five
7x7 box filters run through torch's conv2d on whatever device the images are on (metric plumbing, not the hot path).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SSIMLoss(nn.Module):
    def __init__(self, win_size: int = 7, k1: float = 0.01, k2: float = 0.03):
        super().__init__()
        self.win_size = win_size
        self.k1, self.k2 = k1, k2
        self.register_buffer("w", torch.ones(1, 1, win_size, win_size) / win_size ** 2)
        NP = win_size ** 2
        self.cov_norm = NP / (NP - 1)

    def forward(self, X: torch.Tensor, Y: torch.Tensor, data_range: torch.Tensor):
        w = self.w.to(X)
        data_range = data_range[:, None, None, None]
        C1 = (self.k1 * data_range) ** 2
        C2 = (self.k2 * data_range) ** 2
        ux, uy = F.conv2d(X, w), F.conv2d(Y, w)
        uxx, uyy, uxy = F.conv2d(X * X, w), F.conv2d(Y * Y, w), F.conv2d(X * Y, w)
        vx = self.cov_norm * (uxx - ux * ux)
        vy = self.cov_norm * (uyy - uy * uy)
        vxy = self.cov_norm * (uxy - ux * uy)
        A1, A2, B1, B2 = (2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2)
        S = (A1 * A2) / (B1 * B2)
        return 1 - S.mean()
