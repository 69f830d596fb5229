"""Round 4: pmc_r02.py + the dominant kernels of the other BASELINE configurations (E2EVN's U-Net convolution, the qRIM's 128-channel
convolution, the bf16 training tape's layer / cell / weight-gradient kernels, the general-mask gradient without y), so that their roofline
records carry counter traffic too.
Three launches of every kernel of the headline loop at 1 x 15 x 640 x 372 x 64 features, for the rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, one counter per pass; counter collection costs ~0.3 s per dispatch on this pool, so the bench itself
is out of reach).  tools/traffic_json.py turns the two CSVs into profiles/rNN_traffic.json, which bench.py reports as `traffic`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
B, C, H, W, F = 1, 15, 640, 372, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = r(B, F, H, W), r(B, F, H, W)
wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
pk = ops.rim_layer_wino_pack(wc, wi)
pk1 = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, wi)
eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
mask = (torch.rand(1, 1, 1, W, 1) < 0.3).to(dev)
mask2d = (torch.rand(1, 1, H, W, 1) < 0.1).to(dev)
yt = ops.llg_prepare(y, False, "backward")
op = ops.llg372_prepare(yt, S, mask, False)
wf = r(2, F, 3, 3) / 24
pk2 = ops.rim_layer2_sb_pack(wc, wi, wf)
pk2h = ops.rim_layer2_f16_pack(wc, wi, wf)
xmax = torch.zeros(1, device=dev)
taps = torch.empty(B, 18, H, W, device=dev)
work = torch.empty_like(y)
hpc = ops.cb8_from_nchw(hp)
torch.cuda.synchronize()
for _ in range(3):
    part, n = ops.llg372(eta, op, 1.0, "backward", parts=True)
    h1 = ops.rim_layer1_cb8(None, eta, part, n, 1.0, pk1, bc, bi, hh, hpc, xmax)                          # (keeps the bound of its outputs in xmax)
    ops.rim_layer2_f16_cb8(h1, pk2h, bc, bi, hh, hpc, xmax, taps=taps, want_taps=True)                  # the headline loop's form (channel-blocked states)
    ops.rim_layer_indrnn_wino(x, pk, F, bc, bi, hh, hp)
    ops.rim_layer2_sb_taps(x, pk2, bc, bi, hh, hp, taps)      # the three-term bf16 form (MRIDC_AMD_ARITH=bf16x3)
    ops.rim_final_gather(taps, None, eta)
    ops.llg372_gather(eta, taps, None, op, 1.0, "backward")   # the same gather folded into the next step's gradient launch (the loop's default)
    ops.rim_final(x, wf, None, 3, 1, eta)
# ---- the other configurations' dominant kernels (their own shapes) ------------------------------------------------------------------------------
A14 = r(4, 14, 640, 380)      # (the NormUnet pads 372 to 380 = ((372 - 1) | 11) + 1)
nA = torch.stack([A14.mean((2, 3)), 1 / torch.sqrt(A14.var((2, 3), unbiased=False) + 1e-5)], -1)
W14 = r(14, 14, 3, 3) / 11
X128, W128, B128 = r(1, 128, 256, 256), r(128, 128, 3, 3) / 34, r(128) * 0.1
bnd = ops.max_abs(X128).reshape(1)
cw2, cb2 = r(F, F, 3, 3) / 24, r(F) * 0.1
wih, bih, hh2, wfin = r(F, F, 1, 1) / 8, r(F) * 0.1, r(1, F, 1, 1) * 0.5, r(2, F, 3, 3) / 24
dhP, aP = ops.f32_to_pairs(r(B, F, H, W)), ops.f32_to_pairs(r(B, F, H, W).relu())
dH, hst, xcb = r(B, 8, H, W, 8), torch.randint(-2 ** 31, 2 ** 31 - 1, (B, H, W, 2), dtype=torch.int32, device=dev), ops.cb8_from_nchw(x)     # (the training tape: hidden states channel-blocked, the cell's own state as its (h > 0) mask words)
part_c = ops.tl_cell_part(B, H, W, dev)
torch.cuda.synchronize()
for i in range(3):
    ops.unet_conv3x3((A14, nA), None, W14)                                                       # E2EVN: k_uconv_h<1, 1, true> at batch 4
    ops.conv3x3_h(X128, W128, B128, 2, ops.PAD_REPLICATE, ops.ACT_RELU, bound=bnd)              # qCIRIM: k_uconv_h<4, 2, false>
    ops.tl_layer_fwd(xcb, cw2, cb2, wih, bih, hh2, hpc, wfin)                                       # training: fused second layer forward (+ tap products)
    _, ga = ops.tl_cell_bwd(dhP, dH, hst, hpc, aP, wih, wfin, hh2, part_c, i == 0)                # training: one-pass cell backward
    ops.conv_wgrad_bf16_pairs(xcb, ga, 3, 2, ops.PAD_REPLICATE)                                    # training: 3x3 d2 weight gradient from the pair tensor
    ops.tl_dgrad(ga, cw2, 2, True)                                                               # training: data gradient pairs -> pairs
    ops.llg(eta, y, S, mask2d, 1.0, False, "backward", work=work, parts=True)                    # general mask, deferred form: the column pass without y
torch.cuda.synchronize()
print("done")
