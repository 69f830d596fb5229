"""SSIM loss -- drop-in for `mridc.collections.common.losses.ssim.SSIMLoss` (reference ssim.py:11-61), HIP backed (inference /
metric use: no autograd)."""
import torch
import torch.nn as nn

from mridc_amd import _lib


class SSIMLoss(nn.Module):
    def __init__(self, win_size: int = 7, k1: float = 0.01, k2: float = 0.03):
        super().__init__()
        self.win_size = win_size
        self.k1, self.k2 = k1, k2
        self.register_buffer("w", torch.ones(1, 1, win_size, win_size) / win_size ** 2)
        NP = win_size ** 2
        self.cov_norm = NP / (NP - 1)

    def forward(self, X: torch.Tensor, Y: torch.Tensor, data_range: torch.Tensor):
        """X, Y: [B,1,h,w]; data_range: [B].  Returns 1 - mean(SSIM map) (a 0-d tensor)."""
        if X.dim() != 4 or X.shape[1] != 1 or X.shape != Y.shape:
            raise ValueError(f"SSIMLoss expects X and Y of shape [B,1,h,w], got {tuple(X.shape)} and {tuple(Y.shape)}")
        X, Y, dr = _lib.f32c(X), _lib.f32c(Y), _lib.f32c(data_range.reshape(-1))
        B, _, h, w = [int(v) for v in X.shape]
        if dr.numel() != B:
            dr = dr.expand(B).contiguous()
        L = _lib.lib()
        out = torch.empty(1, dtype=torch.float32, device=X.device)
        work = torch.empty(int(L.mrx_ssim_work_floats(B, h, w)), dtype=torch.float32, device=X.device)
        _lib.check(L.mrx_ssim_loss(_lib.ptr(X), _lib.ptr(Y), _lib.ptr(dr), _lib.ptr(out), _lib.ptr(work), B, h, w, self.win_size,
                                   float(self.k1), float(self.k2), _lib.stream_ptr()), "mrx_ssim_loss")
        return out[0]
