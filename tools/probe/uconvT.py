"""mrx_unet_conv_transpose2x2 at the E2EVN NormUnet shapes, batch 4: us per launch (graph replay), HBM fraction on its algorithmic bytes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timed(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g_, stream=st):
            for _ in range(n):
                fn()
    g_.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g_.replay()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / (n * reps)


B = 4
for Cin, Cout, H, W in [(56, 28, 160, 96), (28, 14, 320, 192), (288, 144, 40, 24), (144, 72, 80, 48), (72, 36, 160, 96), (36, 18, 320, 192)]:
    a = r(B, Cin, H, W)
    na = torch.stack([a.mean((2, 3)), 1.0 / torch.sqrt(a.var((2, 3), unbiased=False) + 1e-5)], -1)
    w = r(Cin, Cout, 2, 2) / (4 * Cin) ** 0.5
    t = timed(lambda: ops.unet_conv_transpose2x2((a, na), w))
    mb = (Cin + 4 * Cout) * H * W * B * 4 / 1e6
    print("%3d -> %3d @%dx%d: %.1f us (%.2f of 8 TB/s)" % (Cin, Cout, H, W, t, mb / t / 8.0), flush=True)
