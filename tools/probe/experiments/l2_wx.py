"""Second RIM layer, Winograd F(2, 3)-along-x form (mrx_rim_layer2_wx_cb8) against the direct two-term fp16 kernel and float64: error at small shapes (odd sizes
included), then time at the bench's launch shape (8 slices of 640 x 372)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
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


w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk_d, pk_w = ops.rim_layer2_f16_pack(w2, wi2, wf), ops.rim_layer2_wx_pack(w2, wi2, wf)
rel = lambda a, b: float((a.double() - b).norm() / b.norm())  # noqa: E731
for (B, H, W) in ((1, 96, 80), (2, 37, 45), (1, 16, 32), (1, 5, 7), (1, 64, 372)):
    x, hp = r(B, F, H, W).relu() * 3.0, r(B, F, H, W).relu()
    xm = x.abs().max().reshape(1).contiguous()
    gd = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), w2.double(), bc.double(), dilation=2).relu()
    ref = Fn.relu(Fn.conv2d(gd, wi2.double(), bi.double()) + hh.double() * hp.double())
    tref = Fn.conv2d(ref, wf.double().permute(2, 3, 0, 1).reshape(18, F, 1, 1))
    xc, hc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
    d_h, d_t = ops.rim_layer2_f16_cb8(xc, pk_d, bc, bi, hh, hc, xm, want_taps=True)
    w_h, w_t = ops.rim_layer2_wx_cb8(xc, pk_w, bc, bi, hh, hc, xm, want_taps=True)
    w0 = ops.rim_layer2_wx_cb8(xc, pk_w, bc, bi, hh, None, xm)
    ref0 = Fn.relu(Fn.conv2d(gd, wi2.double(), bi.double()))
    print(f"{B}x{H}x{W}: h direct {rel(ops.cb8_to_nchw(d_h), ref):.2e}  winograd {rel(ops.cb8_to_nchw(w_h), ref):.2e}   taps direct {rel(d_t, tref):.2e}  winograd {rel(w_t, tref):.2e}"
          f"   zero state, no taps {rel(ops.cb8_to_nchw(w0), ref0):.2e}", flush=True)
B, H, W = int(os.environ.get("PROBE_B", "8")), 640, 372
h1, hpb = ops.cb8_from_nchw(r(B, F, H, W).relu()), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm1 = h1.abs().max().reshape(1).contiguous()
o2, tp = torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
for rep in range(2):
    td = timed(lambda: ops.rim_layer2_f16_cb8(h1, pk_d, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True))
    tw = timed(lambda: ops.rim_layer2_wx_cb8(h1, pk_w, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True))
    print(f"time per slice ({B} per launch): direct {td / B:.2f} us   winograd-x {tw / B:.2f} us", flush=True)
