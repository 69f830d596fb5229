"""mrx_unet_conv3x3_h at the E2EVN shapes under the kernel's phase ablations (probe build: MRX_BUILD_DEFS=-DMRX_PROBE, env MRX_UCONVH_ABLATE:
1 no matrix work, 2 no stores, 4 no statistics, 8 no tile loads, 16 no split / LDS writes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
B = int(os.environ.get("PROBE_B", "8"))
out = []
for Ca, Cout, H, W in [(14, 14, 640, 380), (28, 28, 320, 190)]:
    a = r(B, Ca, H, W)
    na = torch.stack([a.mean((2, 3)), 1.0 / torch.sqrt(a.var((2, 3), unbiased=False) + 1e-5)], -1)
    w = r(Cout, Ca, 3, 3) / (9 * Ca) ** 0.5
    for _ in range(5):
        ops.unet_conv3x3((a, na), None, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.unet_conv3x3((a, na), None, w)
    e.record()
    torch.cuda.synchronize()
    out.append("%d->%d @%dx%d: %.1f us" % (Ca, Cout, H, W, 20 * s.elapsed_time(e)))
print("ablate", os.environ.get("MRX_UCONVH_ABLATE", "0"), " | ".join(out), "(conv + finalize, eager back to back)")
