import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oracle, torch.nn.functional as F
from oracle import amp
from mridc_amd import synthetic
from mridc_amd.collections.reconstruction.models.cirim import CIRIM
cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)
torch.manual_seed(0); model = CIRIM(cfg)
state = {k: v.detach().clone() for k, v in model.state_dict().items()}
s = synthetic.make_slice(15, 640, 372, slice_idx=0)
refs = torch.load("/tmp/seed0_slice0_refs.pt")
# the same arithmetic with fp32 accumulation (a different order of sums, as on the GPU): how far does the gradient move?
class C32(amp._ConvBf16Operands):
    @staticmethod
    def forward(ctx, x, w, b, padding, dilation, round_forward, round_results):
        ctx.save_for_backward(x, w); ctx.cfg = (padding, dilation, b is not None, round_results)
        y = F.conv2d(amp.bf16_round(x), amp.bf16_round(w), None, padding=padding, dilation=dilation)
        return amp.bf16_round(y + amp.bf16_round(b).view(1, -1, 1, 1) if b is not None else y)
amp._ConvBf16Operands.forward = C32.forward
l, g = amp.cirim_loss_and_gradients(state, cfg, s, "bf16_operands", round_results=True)
fl = lambda g: torch.cat([g[k].reshape(-1).double() for k in sorted(g) if not k.endswith("dc_weight")])
a, b = fl(g), fl(refs["oprr"])
print("fp32-accumulated forward vs fp64-accumulated forward (same arithmetic otherwise): whole %.3e" % float((a-b).norm()/b.norm()))
# and with a tie-free target
with torch.no_grad():
    pred = oracle.models.cirim_forward(state, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
for margin in (1e-3, 3e-3):
    s2 = dict(s); s2["target"] = amp.detie_l1_target(s["target"], pred, margin)
    print("margin", margin, "moved", int((s2["target"] != s["target"]).sum()), "of", s["target"].numel())
    amp._ConvBf16Operands.forward = C32.forward
    l1, g1 = amp.cirim_loss_and_gradients(state, cfg, s2, "bf16_operands", round_results=True)
    import importlib
    # fp64 accumulation again
    def fwd64(ctx, x, w, b, padding, dilation, round_forward, round_results):
        ctx.save_for_backward(x, w); ctx.cfg = (padding, dilation, b is not None, round_results)
        y = F.conv2d(amp.bf16_round(x).double(), amp.bf16_round(w).double(), None, padding=padding, dilation=dilation).float()
        return amp.bf16_round(y + amp.bf16_round(b).view(1, -1, 1, 1) if b is not None else y)
    amp._ConvBf16Operands.forward = staticmethod(fwd64)
    l2, g2 = amp.cirim_loss_and_gradients(state, cfg, s2, "bf16_operands", round_results=True)
    a, b = fl(g1), fl(g2)
    print("   tie-free target: fp32- vs fp64-accumulated forward: whole %.3e" % float((a-b).norm()/b.norm()), flush=True)
