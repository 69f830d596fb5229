"""mrx_rim_layer2_cb8 phase ablation (library built with MRX_BUILD_DEFS=-DMRX_PROBE).  MRX_L2C8_ABL bits: 1 no tail slices, 2 no staging
slices, 4 no convolution MFMAs, 8 no operand fetches, 16 both wave halves in the early order, 32 no barriers."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, 372
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = r(B, F, H, W).relu(), r(B, F, H, W).relu()
wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
wf = r(2, F, 3, 3) / 24
pk = ops.rim_layer2_f16_pack(wc, wi, wf)
xmax = x.abs().max().reshape(1).contiguous()
xc, hpc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
taps = torch.empty(B, 9, H, W, 2, device=dev)
out = torch.empty_like(xc)
fn = lambda: ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, taps=taps, out=out, want_taps=True)  # noqa: E731


def timed(n=100):
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


names = {0: "full", 1: "no tail", 2: "no staging", 3: "no tail, no staging", 4: "no conv MFMAs", 7: "no tail / staging / MFMAs (fetches + barriers)",
         11: "no tail / staging / fetches (MFMAs + barriers)", 16: "both halves early", 19: "both early, no tail, no staging", 32: "no barriers",
         35: "no tail, no staging, no barriers"}
for rep in range(2):
    for abl, name in names.items():
        if abl:
            os.environ["MRX_L2C8_ABL"] = str(abl)
        else:
            os.environ.pop("MRX_L2C8_ABL", None)
        t = timed()
        os.environ["MRX_L2C8_TRACE"] = "1"
        sys.stderr.write("ABL %2d %-50s %.2f us   " % (abl, name, t))
        sys.stderr.flush()
        fn()
        torch.cuda.synchronize()
        os.environ.pop("MRX_L2C8_TRACE", None)
