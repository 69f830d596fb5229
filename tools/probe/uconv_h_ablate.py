"""Phases of k_uconv_h by switching them off (probe build: MRX_BUILD_DEFS=-DMRX_PROBE python -m mridc_amd._build)."""
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
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n * reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / (n * reps)


B = 4
for Ca, Cb, Cout, H, W in [(14, 0, 14, 640, 384), (28, 28, 28, 320, 192), (18, 18, 18, 640, 384)]:
    a = r(B, Ca, H, W)
    na = torch.stack([a.mean((2, 3)), 1.0 / torch.sqrt(a.var((2, 3), unbiased=False) + 1e-5)], -1)
    b = r(B, Cb, H, W) if Cb else None
    nb = torch.stack([b.mean((2, 3)), 1.0 / torch.sqrt(b.var((2, 3), unbiased=False) + 1e-5)], -1) if Cb else None
    w = r(Cout, Ca + Cb, 3, 3) / (9 * (Ca + Cb)) ** 0.5
    src_a, src_b = (a, na), ((b, nb) if Cb else None)
    for abl, name in ((0, "full"), (1, "no MFMA"), (2, "no stores"), (4, "no stats"), (8, "no tile loads"), (16, "no LDS writes"), (24, "no staging"),
                      (7, "staging only"), (31, "nothing")):
        os.environ["MRX_UCONVH_ABLATE"] = str(abl)
        print("%d+%d->%d @%dx%d  %-14s %.1f us" % (Ca, Cb, Cout, H, W, name, timed(lambda: ops.unet_conv3x3(src_a, src_b, w))), flush=True)
