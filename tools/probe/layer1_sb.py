"""First RIM layer: the split-bf16 kernel (k_rim_layer1_sb) against the fp32-MFMA kernel (MRIDC_AMD_ARITH=fp32) and a float64 torch reference at
1 x 640 x 372 -- error of both and time per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
eta, part, hp = r(B, H, W, 2), r(3, B, H, W, 2), r(B, F, H, W).relu()
wc, wi = r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8
pk1 = ops.rim_layer_pack(wc, wi)
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
fn = lambda: ops.rim_layer_indrnn_packed_llg(eta, part, 3, 1.0, pk1, F, 5, 1, bc, bi, hh, hp)  # noqa: E731


def timed():
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 5 * s.elapsed_time(e)


x = torch.cat([eta.permute(0, 3, 1, 2), part.sum(0).permute(0, 3, 1, 2)], 1).double()
ref = Fn.relu(Fn.conv2d(Fn.pad(x, (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double()))
ref = Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())
for mode in ("0", "1"):
    os.environ["MRIDC_AMD_ARITH"] = "fp32" if mode == "1" else "bf16x3"
    out = fn()
    err = ((out.double() - ref).norm() / ref.norm()).item()
    print("fp32-MFMA=%s: rel-L2 vs float64 %.3e, max abs %.3e, %.2f us per launch" % (mode, err, (out.double() - ref).abs().max().item(), timed()))
