import sys, torch
sys.path.insert(0, '/root/repo')
from mridc_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B, C, H = 1, 15, 640
for W in (372, 368, 384, 320, 256, 360, 400):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
    mask = (torch.rand(1, 1, 1, W, 1) < 0.3).to(dev)
    out = torch.empty(B, 4, H, W, device=dev)
    yt = ops.llg_prepare(y, False, "backward")
    t = timeit(lambda: ops.llg_hinv(eta, yt, S, mask, 1.0, False, "backward", out=out))
    print(f"W={W}: llg_hinv {t:.1f} us  {(25 + 16 * C) * H * W * B / t / 1e3:.0f} GB/s algorithmic")
