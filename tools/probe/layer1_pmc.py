"""Three launches of the first RIM layer at 1 x 640 x 372 (for rocprofv3 --pmc passes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, 372
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
eta, part, hp = r(B, H, W, 2), r(3, B, H, W, 2), r(B, F, H, W).relu()
pk1 = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8)
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
torch.cuda.synchronize()
for _ in range(3):
    ops.rim_layer_indrnn_packed_llg(eta, part, 3, 1.0, pk1, F, 5, 1, bc, bi, hh, hp)
torch.cuda.synchronize()
