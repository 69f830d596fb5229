"""bf16-storage tape: are two runs of the same step bit-identical?  (a race in the new kernels would show as run-to-run differences)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import autograd as ag
from mridc_amd import synthetic, training
from mridc_amd.collections.reconstruction.models.cirim import CIRIM
C, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 48, 40)
dev = torch.device("cuda:0")
cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
torch.manual_seed(0)
model = CIRIM(cfg).to(dev).train()
s = synthetic.make_slice(C, H, W, slice_idx=2)
batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
ag.set_precision("bf16")
runs = []
for r in range(4):
    for p in model.parameters():
        p.grad = None
    loss = training.cirim_forward_backward(model, batch, "bf16")
    runs.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
for r in range(1, 4):
    bad = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[r][1][n])]
    print(f"run {r} vs run 0: loss {'==' if runs[r][0] == runs[0][0] else '!='}, tensors that differ: {bad if bad else 'none'}")
    for n in bad:
        a, b = runs[0][1][n].double(), runs[r][1][n].double()
        print(f"     {n}: rel {float((a - b).norm() / a.norm()):.3e}")
