"""The two RIM layers on channel-blocked states (mrx_rim_layer1_cb8, mrx_rim_layer2_f16_cb8) against the NCHW entry points: bit-identity, time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, int(os.environ.get("PROBE_H", "640")), int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


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


x4, hp = r(B, 4, H, W), r(B, F, H, W).relu()
wc, wi = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk1 = ops.rim_layer_pack(wc, wi)
xm1, xm2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
ref1 = ops.rim_layer_indrnn_packed(x4, pk1, F, 5, 1, bc, bi, hh, hp, xmax=xm1)
hpc = ops.cb8_from_nchw(hp)
got1 = ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, hpc, xm2)
print("layer 1 cb8 == nchw:", bool(torch.equal(ops.cb8_to_nchw(got1), ref1)), "zero state:",
      bool(torch.equal(ops.cb8_to_nchw(ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, None, xm2)),
                       ops.rim_layer_indrnn_packed(x4, pk1, F, 5, 1, bc, bi, hh, None, xmax=xm1))), flush=True)
x = ref1
w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
pk2 = ops.rim_layer2_f16_pack(w2, wi2, wf)
ref2, t_ref = ops.rim_layer2_f16(x, pk2, bc, bi, hh, hp, xm1, want_taps=True)
got2, t_got = ops.rim_layer2_f16_cb8(got1, pk2, bc, bi, hh, hpc, xm2, want_taps=True)
print("layer 2 cb8 == nchw:", bool(torch.equal(ops.cb8_to_nchw(got2), ref2)), "taps:", bool(torch.equal(t_got, t_ref)), "zero state:",
      bool(torch.equal(ops.cb8_to_nchw(ops.rim_layer2_f16_cb8(got1, pk2, bc, bi, hh, None, xm2)), ops.rim_layer2_f16(x, pk2, bc, bi, hh, None, xm1))), flush=True)
o1, o1c, o2, o2c = torch.empty_like(ref1), torch.empty_like(got1), torch.empty_like(ref2), torch.empty_like(got2)
for rep in range(3):
    print("layer 1: nchw %.2f us, cb8 %.2f us   layer 2 (+ taps): nchw %.2f us, cb8 %.2f us" % (
        timed(lambda: ops.rim_layer_indrnn_packed(x4, pk1, F, 5, 1, bc, bi, hh, hp, out=o1, xmax=xm1)),
        timed(lambda: ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, hpc, xm2, out=o1c)),
        timed(lambda: ops.rim_layer2_f16(x, pk2, bc, bi, hh, hp, xm1, taps=t_ref, out=o2, want_taps=True)),
        timed(lambda: ops.rim_layer2_f16_cb8(got1, pk2, bc, bi, hh, hpc, xm2, taps=t_got, out=o2c, want_taps=True))), flush=True)
