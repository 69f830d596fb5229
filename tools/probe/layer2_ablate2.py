"""Second RIM layer (two-term fp16 route), tail ablations + the per-tile cycle stamps at small grids (one workgroup alone has the HBM to itself).
ABL bits: 128 no h_prev loads, 256 no stores, 512 no tap stage (library built with MRX_BUILD_DEFS=-DMRX_PROBE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")


def setup(H, W):
    g = torch.Generator().manual_seed(0)
    B, F = 1, 64
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(B, F, H, W).relu(), r(B, F, H, W).relu()
    wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
    bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
    wf = r(2, F, 3, 3) / 24
    pk_h = ops.rim_layer2_f16_pack(wc, wi, wf)
    xmax = x.abs().max().reshape(1).contiguous()
    taps = torch.empty(B, 18, H, W, device=dev)
    out = torch.empty_like(hp)
    return lambda: ops.rim_layer2_f16(x, pk_h, bc, bi, hh, hp, xmax, taps=taps, out=out, want_taps=True)


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


if __name__ == "__main__":
    names = {0: "full", 16: "no tail", 128: "no h_prev loads", 256: "no stores", 384: "no h_prev loads, no stores", 512: "no tap stage",
             896: "no loads, stores, tap stage", 903: "no staging, no tail memory ops, no tap stage"}
    fn = setup(640, 372)
    for rep in range(2):
        for abl, name in names.items():
            if abl:
                os.environ["MRX_L2_ABL"] = str(abl)
            else:
                os.environ.pop("MRX_L2_ABL", None)
            print("ABL %3d %-44s %.2f us" % (abl, name, timed(fn)), flush=True)
    os.environ["MRX_L2SB_TRACE"] = "1"
    for abl, name in names.items():
        if abl == 16:
            continue
        if abl:
            os.environ["MRX_L2_ABL"] = str(abl)
        else:
            os.environ.pop("MRX_L2_ABL", None)
        print("trace ABL %d %s" % (abl, name), flush=True)
        fn()
        torch.cuda.synchronize()
    os.environ.pop("MRX_L2_ABL", None)
    for H, W in ((32, 32), (32, 64), (64, 128), (128, 256), (256, 256), (512, 256)):      # 2 tiles on ONE workgroup ... 2 tiles on each of 256
        f = setup(H, W)
        os.environ.pop("MRX_L2SB_TRACE", None)
        f()
        torch.cuda.synchronize()
        os.environ["MRX_L2SB_TRACE"] = "1"
        print("trace %d x %d (%d tiles)" % (H, W, (H // 16) * (W // 32)), flush=True)
        f()
        torch.cuda.synchronize()
