"""Second RIM layer on channel-blocked states (the default route): price of each phase by switching it off (probe build: MRX_BUILD_DEFS=-DMRX_PROBE).
ABL bits: 1 no x loads, 2 no operand split, 4 no LDS staging writes, 8 no conv MFMAs, 16 no tail, 32 no LDS operand reads, 64 no barriers,
128 no h_prev loads, 256 no stores, 512 no tap stage."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, 372
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = ops.cb8_from_nchw(r(B, F, H, W).relu()), ops.cb8_from_nchw(r(B, F, H, W).relu())
wc, wi, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk = ops.rim_layer2_f16_pack(wc, wi, wf)
xmax = x.abs().max().reshape(1).contiguous()
taps = torch.empty(B, 18, H, W, device=dev)
out = torch.empty_like(hp)


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


names = {0: "full", 1: "no x loads", 3: "no x loads, no split", 7: "no staging at all", 8: "no conv MFMAs", 16: "no tail", 32: "no LDS operand reads",
         64: "no barriers", 128: "no h_prev loads", 256: "no stores", 384: "no h_prev loads, no stores", 512: "no tap stage", 896: "no loads, stores, tap stage"}
fn = lambda: ops.rim_layer2_f16_cb8(x, pk, bc, bi, hh, hp, xmax, taps=taps, out=out, want_taps=True)  # noqa: E731
for rep in range(2):
    for abl, name in names.items():
        if abl:
            os.environ["MRX_L2_ABL"] = str(abl)
        else:
            os.environ.pop("MRX_L2_ABL", None)
        print("ABL %3d %-30s %.2f us" % (abl, name, timed(fn)), flush=True)
os.environ["MRX_L2SB_TRACE"] = "1"
for abl, name in names.items():
    if abl:
        os.environ["MRX_L2_ABL"] = str(abl)
    else:
        os.environ.pop("MRX_L2_ABL", None)
    print("trace ABL %d %s" % (abl, name), flush=True)
    fn()
    torch.cuda.synchronize()
