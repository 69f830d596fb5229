"""One line per process: the two precision-16 RIM layer kernels (mrx_amp16_layer1 / _layer2, csrc/rim_amp16.hip) at the bench's launch shape -- 8 slices of
640 x 372 -- by HIP events, with the bytes each launch has to move and the HBM fraction that makes, next to the fp32-class kernels on the same box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
F = 64


def timed(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


wc1, wi1 = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
B, H, W = int(os.environ.get("PROBE_B", "8")), 640, 372
N = H * W
eta, part = r(B, H, W, 2), r(4, B, H, W, 2)
a1, a2 = ops.amp16_layer1_pack(wc1, wi1), ops.amp16_layer2_pack(w2, wi2, wf)
hp16 = ops.amp16_from_nchw(r(B, F, H, W).relu())
h1 = ops.amp16_layer1(None, eta, part, 4, 1.0, a1, bc, bi, hh, hp16)
o1, o2 = torch.empty_like(h1), torch.empty_like(h1)
_, tq, te = ops.amp16_layer2(h1, a2, bc, bi, hh, hp16)
b1 = (2 * F * 2 + 5 * 8) * N * B          # h_prev in + h out (fp16) + eta and four partial planes (complex fp32)
b2 = (3 * F * 2 + 6 * 4) * N * B          # x, h_prev in + h out (fp16) + six tap planes (fp32)
lib = os.path.basename(os.path.dirname(os.environ.get("MRIDC_AMD_LIB", "mridc_amd/lib/x")))


def one(tag):
    t1 = timed(lambda: ops.amp16_layer1(None, eta, part, 4, 1.0, a1, bc, bi, hh, hp16, out=o1))
    t2 = timed(lambda: ops.amp16_layer2(h1, a2, bc, bi, hh, hp16, taps_q=tq, edges=te, out=o2))
    t2n = timed(lambda: ops.amp16_layer2(h1, a2, bc, bi, hh, hp16, out=o2, want_taps=False))
    return (f"{lib:12s} {tag:10s} layer1 {t1 / B:6.2f} us/slice ({b1 / t1 / 1e3:5.0f} GB/s = {b1 / t1 / 1e3 / 8000:.2f} of HBM)   layer2 {t2 / B:6.2f} us/slice "
            f"({b2 / t2 / 1e3:5.0f} GB/s = {b2 / t2 / 1e3 / 8000:.2f} of HBM; without tap planes {t2n / B:6.2f})")


line = one("product")
# probe builds (-DMRX_PROBE): phases switched off -- 1 no input loads, 2 no h_prev loads, 4 no state stores, 8 no tap stage, 16 no convolution MFMAs
for abl in [int(v) for v in os.environ.get("PROBE_ABL", "").split(",") if v]:
    os.environ["MRX_AMP_ABL"] = str(abl)
    line += "\n" + one(f"abl {abl}")
os.environ.pop("MRX_AMP_ABL", None)
if os.environ.get("PROBE_FP32", "1") == "1":
    pk1, pk2 = ops.rim_layer_pack(wc1, wi1), ops.rim_layer2_f16_pack(w2, wi2, wf)
    hpb = ops.cb8_from_nchw(r(B, F, H, W).relu())
    xm1 = torch.zeros(1, device=dev)
    f1 = ops.rim_layer1_cb8(None, eta, part, 4, 1.0, pk1, bc, bi, hh, hpb, xm1)
    p1, p2 = torch.empty_like(f1), torch.empty_like(f1)
    s1 = timed(lambda: ops.rim_layer1_cb8(None, eta, part, 4, 1.0, pk1, bc, bi, hh, hpb, xm1, out=p1))
    s2 = timed(lambda: ops.rim_layer2_f16_cb8_q(f1, pk2, bc, bi, hh, hpb, xm1, out=p2))
    line += f"\n{lib:12s} fp32-class route: layer1 {s1 / B:6.2f}  layer2 {s2 / B:6.2f} us/slice"
print(line, flush=True)
