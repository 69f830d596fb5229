"""Layer 2 with the tap products pre-summed along x (mrx_rim_layer2_f16_cb8_q + mrx_rim_final_gather_q) against the 18-plane route and float64; then time."""
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


w2, wi2, wf, bf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24, r(2) * 0.1
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk = ops.rim_layer2_f16_pack(w2, wi2, wf)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
for (B, H, W) in ((1, 96, 80), (2, 37, 45), (1, 16, 32), (1, 5, 7), (1, 64, 372), (1, 33, 65), (1, 20, 64)):
    x, hp, eta = r(B, F, H, W).relu() * 3.0, r(B, F, H, W).relu(), r(B, H, W, 2)
    xm = x.abs().max().reshape(1).contiguous()
    gd = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), w2.double(), bc.double(), dilation=2).relu()
    ref = Fn.relu(Fn.conv2d(gd, wi2.double(), bi.double()) + hh.double() * hp.double())
    ref_eta = eta.double() + (Fn.conv2d(Fn.pad(ref, (1, 1, 1, 1), mode="replicate"), wf.double()) + bf.double().view(1, 2, 1, 1)).permute(0, 2, 3, 1)
    xc, hc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
    d_h, d_t = ops.rim_layer2_f16_cb8(xc, pk, bc, bi, hh, hc, xm, want_taps=True)
    d_eta = ops.rim_final_gather(d_t, bf, eta)
    q_h, q_t, q_e = ops.rim_layer2_f16_cb8_q(xc, pk, bc, bi, hh, hc, xm)
    q_eta = ops.rim_final_gather_q(q_t, q_e, bf, eta)
    print(f"{B}x{H}x{W}: h identical {bool(torch.equal(d_h, q_h))}   eta vs float64: 18-plane {rel(d_eta, ref_eta):.2e}  pre-summed {rel(q_eta, ref_eta):.2e}   between them {rel(q_eta, d_eta):.2e}"
          f"   max |diff| {float((q_eta - d_eta).abs().max()):.2e}", flush=True)
B, H, W = int(os.environ.get("PROBE_B", "8")), 640, 372
h1, hpb = ops.cb8_from_nchw(r(B, F, H, W).relu()), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm1 = h1.abs().max().reshape(1).contiguous()
o2, tp = torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
_, tq, te = ops.rim_layer2_f16_cb8_q(h1, pk, bc, bi, hh, hpb, xm1)
for rep in range(2):
    td = timed(lambda: ops.rim_layer2_f16_cb8(h1, pk, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True))
    tw = timed(lambda: ops.rim_layer2_f16_cb8_q(h1, pk, bc, bi, hh, hpb, xm1, taps_q=tq, edges=te, out=o2))
    print(f"time per slice ({B} per launch): 18 planes {td / B:.2f} us   pre-summed {tw / B:.2f} us", flush=True)
