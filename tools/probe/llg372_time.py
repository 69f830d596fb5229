"""Time the W = 372 gradient kernels on one 15 x 640 x 372 slice (HIP events over back-to-back launches) and compare them."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops

dev = torch.device("cuda:0")
B, C, H, W = 1, int(os.environ.get("C", 15)), int(os.environ.get("H", 640)), 372
g = torch.Generator().manual_seed(0)
y = torch.randn(B, C, H, W, 2, generator=g).to(dev)
S = torch.randn(B, C, H, W, 2, generator=g).to(dev)
eta = torch.randn(B, H, W, 2, generator=g).to(dev)
mask = (torch.rand(1, 1, 1, W, 1, generator=g) < 0.3).to(dev)
y = y * mask
yt = ops.llg_prepare(y, False, "backward")
op = ops.llg372_prepare(yt, S, mask, False)
ref = ops.llg_hinv(eta, yt, S, mask, 1.0, False, "backward")
got = ops.llg372(eta, op, 1.0, "backward")
print("rel diff new vs old:", float((got - ref).norm() / ref.norm()))


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


print("old llg_hinv_parts: %.2f us" % timeit(lambda: ops.llg_hinv_parts(eta, yt, S, mask, 1.0, False, "backward")))
t = timeit(lambda: ops.llg372(eta, op, 1.0, "backward", parts=True))
print("new llg372 (parts): %.2f us -> %.3f of 8 TB/s on %.2f MB" % (t, (25 + 16 * C) * H * W / (t * 1e-6) / 8e12, (25 + 16 * C) * H * W / 1e6))
