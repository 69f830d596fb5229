"""Do the two RIM layers overlap when they CAN share a CU?  Layer 1 (HBM-bound, matrix pipe 0.34 busy) on stream A and layer 2 (matrix-bound, 0.41 of
HBM) on stream B, 8 slices per launch each, N launches per stream: the two streams alone, then together.  With the product kernels every persistent
workgroup owns its CU's whole register file, so `together` ~ `alone A + alone B`; the probe libraries hold the small-footprint forms (layer 2 with four
waves of two rows -- 240 registers per SIMD, 65 KB of LDS; layer 1 with eight waves -- 256 registers per SIMD, 70 KB): one workgroup of each fits a CU."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
F, B, H, W = 64, 8, 640, 372
wc1, wi1 = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk1, pk2 = ops.rim_layer_pack(wc1, wi1), ops.rim_layer2_f16_pack(w2, wi2, wf)
# two independent batches (what the bench's two streams hold)
eta, part, hpa = r(B, H, W, 2), r(4, B, H, W, 2), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm_a, xm_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
h1b, hpb = ops.cb8_from_nchw(r(B, F, H, W).relu()), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm_b.fill_(float(h1b.abs().max()))
o1, o2, tp = torch.empty_like(hpa), torch.empty_like(hpb), torch.empty(B, 18, H, W, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
l1 = lambda: ops.rim_layer1_cb8(None, eta, part, 4, 1.0, pk1, bc, bi, hh, hpa, xm_a, out=o1)  # noqa: E731
l2 = lambda: ops.rim_layer2_f16_cb8(h1b, pk2, bc, bi, hh, hpb, xm_b, taps=tp, out=o2, want_taps=True)  # noqa: E731


def run(fa, fb, n=30):
    for f, st in ((fa, sa), (fb, sb)):
        if f:
            with torch.cuda.stream(st):
                for _ in range(4):
                    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n / B


lib = os.path.basename(os.path.dirname(os.environ.get("MRIDC_AMD_LIB", "mridc_amd/lib/x")))
for rep in range(2):
    ta, tb, tab = run(l1, None), run(None, l2), run(l1, l2)
    print(f"{lib:14s} layer 1 alone {ta:6.2f}  layer 2 alone {tb:6.2f}  sum {ta + tb:6.2f}  together {tab:6.2f} us per slice pair   (together / sum = {tab / (ta + tb):.3f})", flush=True)
