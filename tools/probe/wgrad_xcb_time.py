"""Final convolution's weight gradient of the training tape (3x3, 64 -> 2, x channel-blocked, dy fp32) at 1 x 640 x 372: isolated launch time by HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mridc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 1, 640, 372
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 8, H, W, 8, generator=g).to(dev)
dy = torch.randn(B, 2, H, W, generator=g).to(dev)
out = torch.zeros(2, 64, 3, 3, device=dev)


def t(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lib = os.path.basename(os.path.dirname(os.environ.get("MRIDC_AMD_LIB", "mridc_amd/lib/x")))
print(f"{lib:14s} final-conv wgrad {t(lambda: ops.conv_wgrad_bf16_xcb(x, dy, ops.PAD_REPLICATE, out=out, accumulate=True)):.1f} us (launch + 512-slot reduction)", flush=True)
