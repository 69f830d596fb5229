"""3x3 convolution 64 -> 64 (dilation 1 | 2, replicate padding, ReLU) at 640 x 372: the three-term bf16 kernel of the RIM layer (mrx_conv3x3_sb) against
the two-term fp16 any-channel kernel (mrx_conv3x3_h) with a measured bound (mrx_max_abs) and with a given one."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


x, w, b = r(1, 64, 640, 372).relu(), r(64, 64, 3, 3) / 24, r(64) * 0.1
bound = x.abs().max().reshape(1)
out = torch.empty_like(x)
for dil in (1, 2):
    t_sb = timed(lambda: ops.conv3x3_sb(x, w, b, dil, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0, out))
    t_h = timed(lambda: ops.conv3x3_h(x, w, b, dil, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0, out, bound=bound))
    t_hm = timed(lambda: ops.conv3x3_h(x, w, b, dil, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0, out))
    print("dilation %d: conv3x3_sb (3 x bf16) %.1f us, conv3x3_h (2 x fp16, bound given) %.1f us, with mrx_max_abs %.1f us" % (dil, t_sb, t_h, t_hm), flush=True)
