"""Cycle stamps of the dominant layer (PROBE build: MRX_BUILD_DEFS=-DMRX_PROBE, env MRX_L2SB_TRACE=1): chunk loop / 1x1 stage / rest of the tail per tile,
8 slices of 640 x 372 on channel-blocked states."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
F, B, H, W = 64, 8, 640, 372
w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
pk2 = ops.rim_layer2_f16_pack(w2, wi2, wf)
h1, hpb = ops.cb8_from_nchw(r(B, F, H, W).relu()), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm1 = h1.abs().max().reshape(1).contiguous()
o2, tp = torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
for _ in range(3):
    ops.rim_layer2_f16_cb8(h1, pk2, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True)
torch.cuda.synchronize()
os.environ["MRX_L2SB_TRACE"] = "1"
for _ in range(3):
    ops.rim_layer2_f16_cb8(h1, pk2, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True)
    torch.cuda.synchronize()
