"""mrx_unet_conv3x3_h (two-term fp16) against mrx_unet_conv3x3 (fp32-input MFMA) at the E2EVN NormUnet shapes, batch 4: us per launch, HBM fraction."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timed(fn, n=20, reps=5):
    """us per call, the calls replayed from a hipGraph (no host launch cost)."""
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


B = int(os.environ.get("PROBE_B", "4"))
shapes = [(2, 0, 14, 640, 384), (14, 0, 14, 640, 384), (14, 14, 14, 640, 384), (14, 0, 28, 320, 192), (28, 0, 28, 320, 192), (28, 28, 28, 320, 192),
          (28, 0, 56, 160, 96), (56, 0, 56, 160, 96), (18, 0, 18, 640, 384), (18, 18, 18, 640, 384), (36, 36, 36, 320, 192), (72, 72, 72, 160, 96),
          (144, 144, 144, 80, 48), (144, 0, 288, 40, 24), (288, 0, 288, 40, 24)]
for Ca, Cb, Cout, H, W in shapes:
    a = r(B, Ca, H, W)
    na = torch.stack([a.mean((2, 3)), 1.0 / torch.sqrt(a.var((2, 3), unbiased=False) + 1e-5)], -1)
    b = r(B, Cb, H, W) if Cb else None
    nb = torch.stack([b.mean((2, 3)), 1.0 / torch.sqrt(b.var((2, 3), unbiased=False) + 1e-5)], -1) if Cb else None
    w = r(Cout, Ca + Cb, 3, 3) / (9 * (Ca + Cb)) ** 0.5
    src_a, src_b = (a, na), ((b, nb) if Cb else None)
    ops.UNET_F16 = True
    t16 = timed(lambda: ops.unet_conv3x3(src_a, src_b, w))
    ops.UNET_F16 = False
    t32 = timed(lambda: ops.unet_conv3x3(src_a, src_b, w))
    mb = (Ca + Cb + Cout) * H * W * B * 4 / 1e6
    print("%3d+%3d -> %3d @%dx%d: f16 %.1f us (%.2f of 8 TB/s), fp32 %.1f us" % (Ca, Cb, Cout, H, W, t16, mb / t16 / 8.0, t32), flush=True)
