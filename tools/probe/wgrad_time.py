"""Time the training tape's 3x3 weight gradient (mrx_conv_wgrad_bf16_pairs, x channel-blocked) alone: python tools/probe/wgrad_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
B, H, W = 1, 640, 372
g = torch.Generator().manual_seed(0)
x = ops.cb8_from_nchw(torch.randn(B, 64, H, W, generator=g).to(dev))
dy = ops.f32_to_pairs(torch.randn(B, 64, H, W, generator=g).to(dev))
for _ in range(3):
    ops.conv_wgrad_bf16_pairs(x, dy, 3, 2, ops.PAD_REPLICATE)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv_wgrad_bf16_pairs(x, dy, 3, 2, ops.PAD_REPLICATE)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("MRIDC_AMD_LIB", "product"), "wgrad 3x3 d2 pairs/cb8 + reduce: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
