"""A few launches of the dominant layer (channel-blocked, two-term fp16, 8 slices of 640 x 372) for rocprofv3 --pmc passes: where do its waves spend their
cycles (SQ_WAIT_ANY = parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_* = issuing), what clock does the chip sustain
(GRBM_GUI_ACTIVE / duration)."""
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
x4, hpb = r(B, 4, H, W), ops.cb8_from_nchw(r(B, F, H, W).relu())
xm1 = torch.zeros(1, device=dev)
h1 = ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, hpb, xm1)
o1, o2, tp = torch.empty_like(h1), torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
for _ in range(6):
    ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, hpb, xm1, out=o1)
    ops.rim_layer2_f16_cb8(h1, pk2, bc, bi, hh, hpb, xm1, taps=tp, out=o2, want_taps=True)
torch.cuda.synchronize()
