"""One line per process: the two RIM layer kernels of the library MRIDC_AMD_LIB points at (channel-blocked states, two-term fp16 route) at the bench's
launch shape -- 8 slices of 640 x 372 -- by HIP events, plus layer 2's error against a float64 reference at 1 x 64 x 96 x 80.  Run once per variant
library, alternating (tools/runs/r05*.sh): the boxes differ by a few per cent, the order within a box does not."""
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


wc1, wi1 = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk1, pk2 = ops.rim_layer_pack(wc1, wi1), ops.rim_layer2_f16_pack(w2, wi2, wf)
# accuracy against float64 (small)
x, hp = r(1, F, 96, 80).relu() * 3.0, r(1, F, 96, 80).relu()
xm = x.abs().max().reshape(1).contiguous()
got, taps = ops.rim_layer2_f16_cb8(ops.cb8_from_nchw(x), pk2, bc, bi, hh, ops.cb8_from_nchw(hp), xm, want_taps=True)
gd = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), w2.double(), bc.double(), dilation=2).relu()
ref = Fn.relu(Fn.conv2d(gd, wi2.double(), bi.double()) + hh.double() * hp.double())
tref = Fn.conv2d(ref, wf.double().permute(2, 3, 0, 1).reshape(18, F, 1, 1))
e_h = float((ops.cb8_to_nchw(got).double() - ref).norm() / ref.norm())
e_t = float((taps.double() - tref).norm() / tref.norm())
# time (the bench's launch shape)
B, H, W = int(os.environ.get("PROBE_B", "8")), 640, 372
x4, hpb = r(B, 4, H, W), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm1 = torch.zeros(1, device=dev)
h1 = ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, hpb, xm1)
o1, o2, tp = torch.empty_like(h1), torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
eta, part = r(B, H, W, 2), r(4, B, H, W, 2)             # the headline loop's input form: eta + three coil-group partial planes + the constant plane
t1 = timed(lambda: ops.rim_layer1_cb8(None, eta, part, 4, 1.0, pk1, bc, bi, hh, hpb, xm1, out=o1))
t2 = timed(lambda: ops.rim_layer2_f16_cb8(h1, pk2, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True))
t2b = timed(lambda: ops.rim_layer2_f16_cb8(h1, pk2, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True))
lib = os.path.basename(os.path.dirname(os.environ.get("MRIDC_AMD_LIB", "mridc_amd/lib/x")))
print(f"{lib:12s} layer2 {t2 / B:7.2f} / {t2b / B:7.2f} us per slice ({B} per launch)   layer1 {t1 / B:6.2f}   err vs f64: h {e_h:.2e} taps {e_t:.2e}", flush=True)
