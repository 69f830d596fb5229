"""Second RIM layer: the direct split-bf16 kernel (k_rim_layer2_sb) against the fp32 Winograd kernel and a float64 torch reference at
1 x 64 x 640 x 372 -- error of both and time per launch."""
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
x, hp = r(B, F, H, W).relu(), r(B, F, H, W).relu()
wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk_w = ops.rim_layer_wino_pack(wc, wi)
wf, bf, eta = r(2, F, 3, 3) / 24, r(2) * 0.1, r(B, H, W, 2)
pk_s = ops.rim_layer2_sb_pack(wc, wi, wf)
work = torch.empty(18 * B * H * W, device=dev)
fns = {"winograd fp32": lambda: ops.rim_layer_indrnn_wino(x, pk_w, F, bc, bi, hh, hp),
       "direct split-bf16": lambda: ops.rim_layer2_sb(x, pk_s, bc, bi, hh, hp),
       "final conv alone": lambda: ops.rim_final(hp, wf, bf, 3, 1, eta),
       "split-bf16 + final": lambda: ops.rim_layer2_sb_final(x, pk_s, bc, bi, hh, hp, bf, eta, work)}


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


ref = Fn.relu(Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double(), dilation=2))
ref = Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())
for name, fn in fns.items():
    out = fn()
    if "final" in name:
        print("%-18s %.2f us per call" % (name, timed(fn)))
        continue
    err = ((out.double() - ref).norm() / ref.norm()).item()
    print("%-18s rel-L2 vs float64 %.3e, max abs %.3e, %.2f us per launch" % (name, err, (out.double() - ref).abs().max().item(), timed(fn)))
os.environ["MRX_L2SB_TRACE"] = "1"
fns["direct split-bf16"]()
torch.cuda.synchronize()
