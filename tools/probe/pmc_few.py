"""A handful of launches of each hot kernel at the headline shape (for rocprofv3 --pmc FETCH_SIZE WRITE_SIZE: counter collection costs
~0.3 s per dispatch on this pool, so the full bench is out of reach)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from mridc_amd import ops
dev = torch.device('cuda:0')
B, C, H, W, F = 1, 15, 640, 372, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x, hp = r(B, F, H, W), r(B, F, H, W)
wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
pk = ops.rim_layer_wino_pack(wc, wi)
x4 = r(B, 4, H, W)
pk1 = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, wi)
eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
mask = (torch.rand(1, 1, 1, W, 1) < 0.3).to(dev)
yt = ops.llg_prepare(y, False, "backward")
wf = r(2, F, 3, 3) / 24
torch.cuda.synchronize()
for _ in range(3):
    ops.rim_layer_indrnn_wino(x, pk, F, bc, bi, hh, hp)
    ops.rim_layer_indrnn_packed(x4, pk1, F, 5, 1, bc, bi, hh, hp)
    ops.llg_hinv(eta, yt, S, mask, 1.0, False, "backward")
    ops.rim_final(x, wf, None, 3, 1, eta)
torch.cuda.synchronize()
print("done")
