"""First RIM layer: channel-blocked states (mrx_rim_layer1_cb8) against the NCHW kernel: bit-identity and time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, int(os.environ.get("PROBE_H", "640")), int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = r(B, 4, H, W), r(B, F, H, W).relu()
wc, wi = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk = ops.rim_layer_pack(wc, wi)
xm1, xm2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
ref = ops.rim_layer_indrnn_packed(x, pk, F, 5, 1, bc, bi, hh, hp, xmax=xm1)
hpc = ops.cb8_from_nchw(hp)
got = ops.cb8_to_nchw(ops.rim_layer1_cb8(x, None, None, 0, 1.0, pk, bc, bi, hh, hpc, xm2))
print("cb8 vs nchw: equal", bool(torch.equal(got, ref)), "max", float((got - ref).abs().max()), "xmax", float(xm1), float(xm2))
ref0 = ops.rim_layer_indrnn_packed(x, pk, F, 5, 1, bc, bi, hh, None, xmax=xm1)
got0 = ops.cb8_to_nchw(ops.rim_layer1_cb8(x, None, None, 0, 1.0, pk, bc, bi, hh, None, xm2))
print("zero state: equal", bool(torch.equal(got0, ref0)))


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


o1, o2 = torch.empty_like(ref), torch.empty_like(hpc)
for rep in range(3):
    print("nchw %.2f us (in place %.2f)   cb8 %.2f us (in place %.2f)" % (
        timed(lambda: ops.rim_layer_indrnn_packed(x, pk, F, 5, 1, bc, bi, hh, hp, out=o1, xmax=xm1)),
        timed(lambda: ops.rim_layer_indrnn_packed(x, pk, F, 5, 1, bc, bi, hh, o1, out=o1, xmax=xm1)),
        timed(lambda: ops.rim_layer1_cb8(x, None, None, 0, 1.0, pk, bc, bi, hh, hpc, xm2, out=o2)),
        timed(lambda: ops.rim_layer1_cb8(x, None, None, 0, 1.0, pk, bc, bi, hh, o2, xm2, out=o2))), flush=True)
