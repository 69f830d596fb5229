"""mrx_llg372_prepare (once per slice: S and IFFT_H(y) into the lane order of k_llg372) at 8 x 15 x 640 x 372: launch time by HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import _lib
dev = torch.device("cuda:0")
B, C, H, W = 8, 15, 640, 372
g = torch.Generator().manual_seed(0)
yt, S = torch.randn(B, C, H, W, 2, generator=g).to(dev), torch.randn(B, C, H, W, 2, generator=g).to(dev)
mask = (torch.rand(1, 1, 1, W, 1, generator=g) < 0.3).float().to(dev)
L = _lib.lib()
n = int(L.mrx_llg372_operand_floats(B, C, H))
ytp, Sp, maskp = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.empty(W, device=dev)
m, kind, ms = _lib.mask_args(mask, B, C, H, W)


def call():
    _lib.check(L.mrx_llg372_prepare(_lib.ptr(yt), _lib.ptr(S), _lib.ptr(m), kind, ms, _lib.ptr(ytp), _lib.ptr(Sp), _lib.ptr(maskp), B, C, H, 1, _lib.stream_ptr()), "prep")


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
print(f"mrx_llg372_prepare, {B} slices: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us   checksum {float(ytp.double().sum()):.6f} {float(Sp.double().sum()):.6f} {float((ytp.double() * torch.arange(n, device=dev) % 7).sum()):.3f}")
