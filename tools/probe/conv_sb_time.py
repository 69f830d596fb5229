"""3x3 64 -> 64 convolution at 640 x 372: split-bf16 direct kernel (mrx_conv3x3_sb) against the fp32 Winograd kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, w, b = r(1, 64, 640, 372), r(64, 64, 3, 3) / 24, r(64) * 0.1
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n
for dil in (1, 2):
    for pm in (ops.PAD_ZERO, ops.PAD_REPLICATE):
        ops.SB_CONV = True
        t1 = timeit(lambda: ops.conv2d(x, w, b, dil, pm, ops.ACT_RELU))
        ops.SB_CONV = False
        t0 = timeit(lambda: ops.conv2d(x, w, b, dil, pm, ops.ACT_RELU))
        print(f"dilation {dil} pad {pm}: split-bf16 direct {t1:.1f} us | fp32 Winograd {t0:.1f} us")
