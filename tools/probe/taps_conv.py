import torch, torch.nn.functional as Fn, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/mridc_amd") else os.getcwd())
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n
for (B, Cin, Cout, H, W) in [(1, 128, 4, 256, 256), (1, 64, 2, 640, 372), (2, 64, 3, 37, 29), (1, 128, 2, 33, 47)]:
    x, w, b = r(B, Cin, H, W), r(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5, r(Cout) * 0.1
    for pm, mode in ((ops.PAD_ZERO, "constant"), (ops.PAD_REPLICATE, "replicate")):
        ref = Fn.conv2d(Fn.pad(x.double(), (1, 1, 1, 1), mode=mode), w.double(), b.double())
        ops.TAPS_CONV = True
        got = ops.conv2d(x, w, b, 1, pm)
        t1 = timeit(lambda: ops.conv2d(x, w, b, 1, pm))
        ops.TAPS_CONV = False
        old = ops.conv2d(x, w, b, 1, pm)
        t0 = timeit(lambda: ops.conv2d(x, w, b, 1, pm))
        e = lambda a: float((a.double() - ref).norm() / ref.norm())
        print(f"{Cin}->{Cout} @{H}x{W} pad {mode}: taps rel-L2 {e(got):.2e} {t1:.1f} us | direct {e(old):.2e} {t0:.1f} us")
