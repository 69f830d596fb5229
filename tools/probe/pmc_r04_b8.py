"""The headline loop's kernels at the bench's default batch (8 slices per launch and stream: 3840 tiles = 15 full rounds of the persistent layer
kernels on 256 CUs), three launches each, for the rocprofv3 --pmc passes; tools/traffic_json.py ... 8 tools/probe/pmc_r04_b8.py turns the CSVs into
profiles/r04_traffic_b8.json (shape batch = 8), which bench.py reports as `traffic` / `mfma_util_pmc` of the default line.
`python3 pmc_r04_b8.py 4 2d`: the 2-D-mask line's loop at ITS default batch of 4 (three-pass general-mask gradient instead of mrx_llg372)
-> profiles/r04_traffic_b4.json."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
B, C, H, W, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 15, 640, 372, 64
MASK2D = len(sys.argv) > 2 and sys.argv[2] == "2d"
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
hp = r(B, F, H, W)
wc, wi, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
pk1 = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, wi)
pk2h = ops.rim_layer2_f16_pack(wc, wi, wf)
eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
mask = (torch.rand(1, 1, H, W, 1) < 0.1).to(dev) if MASK2D else (torch.rand(1, 1, 1, W, 1) < 0.3).to(dev)
if not MASK2D:
    yt = ops.llg_prepare(y, False, "backward")
    op = ops.llg372_prepare(yt, S, mask, False)
xmax = torch.zeros(1, device=dev)
taps = torch.empty(B, 18, H, W, device=dev)
hpc = ops.cb8_from_nchw(hp)
torch.cuda.synchronize()
for _ in range(3):
    if MASK2D:
        part, n = ops.llg(eta, y, S, mask, 1.0, False, "backward", parts=True)      # expand / column pass + DC / reduce on the column-tiled coil stack
    else:
        part, n = ops.llg372(eta, op, 1.0, "backward", parts=True)
    h1 = ops.rim_layer1_cb8(None, eta, part, n, 1.0, pk1, bc, bi, hh, hpc, xmax)
    if MASK2D or not ops.RIM_TAPS_Q:
        ops.rim_layer2_f16_cb8(h1, pk2h, bc, bi, hh, hpc, xmax, taps=taps, want_taps=True)
        ops.rim_final_gather(taps, None, eta)
        if not MASK2D:
            ops.llg372_gather(eta, taps, None, op, 1.0, "backward")
    else:                                              # the headline's route since lib 260: tap products pre-summed along x (6 planes + tile-edge terms)
        _, tq, te = ops.rim_layer2_f16_cb8_q(h1, pk2h, bc, bi, hh, hpc, xmax)
        ops.rim_final_gather_q(tq, te, None, eta)
        ops.llg372_gather_q(eta, tq, te, None, op, 1.0, "backward")
torch.cuda.synchronize()
if MASK2D:
    sys.exit(0)
# round 6: the precision-16 route's two layer kernels at the same launch shape (fp16 channel-blocked states; mrx_amp16_layer1 / _layer2)
a1, a2 = ops.amp16_layer1_pack(r(F, 4, 5, 5) / 10, wi), ops.amp16_layer2_pack(wc, wi, wf)
hp16 = ops.amp16_from_nchw(hp.relu())
part4, n4 = ops.llg372(eta, op, 1.0, "backward", parts=True)
torch.cuda.synchronize()
for _ in range(3):
    h16 = ops.amp16_layer1(None, eta, part4, n4, 1.0, a1, bc, bi, hh, hp16)
    ops.amp16_layer2(h16, a2, bc, bi, hh, hp16)
torch.cuda.synchronize()
# E2EVN's dominant launch at the default line's batch: the 14 -> 14 convolution of the first U-Net level, 8 x 640 x 380 (k_uconv_h<1, 1, true>)
A14 = r(8, 14, 640, 380)
nA = torch.stack([A14.mean((2, 3)), 1 / torch.sqrt(A14.var((2, 3), unbiased=False) + 1e-5)], -1)
W14 = r(14, 14, 3, 3) / 11
torch.cuda.synchronize()
for _ in range(3):
    ops.unet_conv3x3((A14, nA), None, W14)
torch.cuda.synchronize()
