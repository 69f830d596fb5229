"""Few-channel convolutions (the cascades' first layers): split-bf16 kernel (mrx_conv_sbs) against the generic / tuned fp32-MFMA kernels."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n
for (Cin, Cout, k, H, W) in [(8, 128, 5, 256, 256), (2, 64, 3, 640, 372), (4, 64, 5, 640, 372), (2, 32, 3, 640, 372)]:
    x, w, b = r(1, Cin, H, W), r(Cout, Cin, k, k) / 10, r(Cout) * 0.1
    ops.SBS_CONV = True
    t1 = timeit(lambda: ops.conv2d(x, w, b, 1, ops.PAD_REPLICATE, ops.ACT_RELU))
    ops.SBS_CONV = False
    t0 = timeit(lambda: ops.conv2d(x, w, b, 1, ops.PAD_REPLICATE, ops.ACT_RELU))
    print(f"{k}x{k} {Cin}->{Cout} @{H}x{W}: split-bf16 {t1:.1f} us | fp32 MFMA {t0:.1f} us")
