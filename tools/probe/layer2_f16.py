"""Second RIM layer: two-term fp16 convolution operands (mrx_rim_layer2_f16) against the three-term bf16 kernel and float64."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, int(os.environ.get("PROBE_H", "640")), int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
for xs, ws in ((1.0, 1.0), (1e-3, 30.0), (3e3, 1e-2)):
    x, hp = r(B, F, H, W).relu() * xs, r(B, F, H, W).relu() * xs
    wc, wi = r(F, F, 3, 3) / 24 * ws, r(F, F, 1, 1) / 8
    bc, bi, hh = r(F) * 0.1 * xs * ws, r(F) * 0.1, r(1, F, 1, 1) * 0.5
    wf = r(2, F, 3, 3) / 24
    pk_s, pk_h = ops.rim_layer2_sb_pack(wc, wi, wf), ops.rim_layer2_f16_pack(wc, wi, wf)
    xmax = x.abs().max().reshape(1).contiguous()
    ref = Fn.relu(Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double(), dilation=2))
    ref = Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())
    fns = {"bf16 x 3": lambda: ops.rim_layer2_sb(x, pk_s, bc, bi, hh, hp), "fp16 x 2": lambda: ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax),
           "fp16 x 2, bound x 1000": lambda: ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax * 1000)}

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            fn()
        e.record()
        torch.cuda.synchronize()
        return 10 * s.elapsed_time(e)

    for name, fn in fns.items():
        out = fn()
        err = ((out.double() - ref).norm() / ref.norm()).item()
        print("scales x %g w %g: %-24s rel-L2 vs float64 %.3e, max abs %.3e, %.2f us" % (xs, ws, name, err, (out.double() - ref).abs().max().item(), timed(fn)))
    h1, t1 = ops.rim_layer2_sb_taps(x, pk_s, bc, bi, hh, hp)
    h2, t2 = ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax, want_taps=True)
    print("   taps: rel diff %.3e; with taps %.2f us (bf16: %.2f us)" % (((t1 - t2).norm() / t1.norm()).item(),
          timed(lambda: ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax, taps=t2, want_taps=True)), timed(lambda: ops.rim_layer2_sb_taps(x, pk_s, bc, bi, hh, hp, t1))))
os.environ["MRX_L2SB_TRACE"] = "1"
ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax)
torch.cuda.synchronize()
