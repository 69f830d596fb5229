"""Race / determinism check: the headline CIRIM step and one training step, repeated; every repetition must reproduce the first
result bit for bit (all reductions in the library have a fixed order; a data race would show up as a changing result)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from mridc_amd import synthetic, training
from mridc_amd.collections.reconstruction.models.cirim import CIRIM

dev = torch.device("cuda:0")
cfg = dict(synthetic.CIRIM_BASELINE_CFG)
cfg["num_cascades"] = 2
torch.manual_seed(0)
model = CIRIM(cfg).to(dev).eval()
s = synthetic.make_slice(15, 640, 372, slice_idx=0)
b = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
ref = None
with torch.no_grad():
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        out = next(model(b["y"], b["sensitivity_maps"], b["mask"], None, b["target"]))[-1][-1]
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            print("MISMATCH at inference repetition", i, float((torch.view_as_real(out) - torch.view_as_real(ref)).abs().max()))
            sys.exit(1)
print("inference: bit-identical over the repetitions")
g0 = None
for i in range(5):
    model.train()
    model.zero_grad(set_to_none=True)
    etas = next(model(b["y"], b["sensitivity_maps"], b["mask"], None, b["target"]))
    loss = training.cirim_l1_loss(etas, b["target"], model.time_steps, len(model.cirim))
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    if g0 is None:
        g0 = g.clone()
    elif not torch.equal(g, g0):
        print("MISMATCH in gradients at repetition", i, float((g - g0).abs().max()))
        sys.exit(1)
print("training: loss", float(loss.detach()), "gradients bit-identical over the repetitions")
