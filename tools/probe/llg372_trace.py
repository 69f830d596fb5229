"""Phase timeline of k_llg372 (MRX_LLG372_ABLATE=3 stamps): one traced launch after warm-up."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MRX_LLG372_ABLATE"] = "3"
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 1, 15, 640, 372
y = torch.randn(B, C, H, W, 2, generator=g).to(dev)
S = torch.randn(B, C, H, W, 2, generator=g).to(dev)
eta = torch.randn(B, H, W, 2, generator=g).to(dev)
mask = (torch.rand(1, 1, 1, W, 1, generator=g) < 0.3).to(dev)
yt = ops.llg_prepare(y * mask, False, "backward")
op = ops.llg372_prepare(yt, S, mask, False)
for _ in range(5):
    ops.llg372(eta, op, 1.0, "backward", parts=True)
torch.cuda.synchronize()
os.environ["MRX_LLG372_TRACE_DUMP"] = "1"
ops.llg372(eta, op, 1.0, "backward", parts=True)
torch.cuda.synchronize()
