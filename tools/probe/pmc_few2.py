"""A handful of launches of the kernels outside the headline loop at 64 x 640 x 372 (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from mridc_amd import ops
dev = torch.device('cuda:0')
B, H, W, F = 1, 640, 372, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x, h = r(B, F, H, W), r(B, F, H, W)
w3, b3 = r(F, F, 3, 3) / 24, r(F)
w1, hh = r(F, F, 1, 1) / 8, r(1, F, 1, 1)
wi, wh, bi = r(3 * F, F, 1, 1) / 8, r(3 * F, F, 1, 1) / 8, r(3 * F)
pk = ops.gated_cell_pack(wi, wh, 3)
w2 = r(2, F, 3, 3) / 24
dy = r(B, F, H, W)
torch.cuda.synchronize()
for _ in range(3):
    ops.conv3x3_wino(x, w3, b3, 1, ops.PAD_ZERO, ops.ACT_RELU)
    ops.conv1x1_64(x, w1, b3, ops.ACT_RELU, 0.0, hh, h)
    ops.gated_cell_1x1(x, h, pk, bi, 3)
    ops.conv_to_complex(x, w2, None, 1, ops.PAD_ZERO)
    ops.conv_wgrad(x, dy, 3, 2, ops.PAD_REPLICATE)
    ops.relu_bwd(dy, x, h, hh)
torch.cuda.synchronize()
print("done")
