"""HIP bf16 training tape against the three oracle arithmetics (oracle/amp.py), per parameter tensor: one cascade at C x H x W.
usage: python tools/probe/train_parity.py [C H W] [precision]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from mridc_amd import autograd as ag
from mridc_amd import synthetic, training
from mridc_amd.collections.reconstruction.models.cirim import CIRIM

C, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (15, 640, 372)
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
quick = "--quick" in sys.argv          # fp32 oracle only, the bench's weights only
ncasc = int(sys.argv[sys.argv.index("--cascades") + 1]) if "--cascades" in sys.argv else 1
slice_override = int(sys.argv[sys.argv.index("--slice") + 1]) if "--slice" in sys.argv else None
dev = torch.device("cuda:0")
# the arithmetic the tape claims: bf16 storage (training.BF16_STORAGE) rounds results too; the round-2 tape keeps the final convolution's forward in fp32
emul = dict(round_results=True) if training.BF16_STORAGE else dict(fp32_forward=((64, 2),))
for seed, boost, sl in (((0, 1.0, 0),) if quick else ((0, 1.0, 0), (5, 3.0, 7))):
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=ncasc)
    sl = sl if slice_override is None else slice_override
    torch.manual_seed(seed)
    model = CIRIM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if boost != 1.0:
                if n_.endswith("bias"):
                    p_.normal_(0, 0.05)
                if n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh"):
                    p_.mul_(boost)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s = synthetic.make_slice(C, H, W, slice_idx=sl)
    refs = {m: oracle.amp.cirim_loss_and_gradients(state, cfg, s, m, **(emul if m == "bf16_operands" else {}))
            for m in (("fp32",) if quick else ("fp32", "autocast_bf16", "bf16_operands"))}
    model = model.to(dev).train()
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    ag.set_precision(prec)
    for prm in model.parameters():
        prm.grad = None
    loss = training.cirim_forward_backward(model, batch, prec)
    ag.set_precision("f32")
    print(f"seed {seed} boost {boost}: loss hip {float(loss):.6f}  " + "  ".join(f"{m} {float(refs[m][0]):.6f}" for m in refs))
    tot = {m: [0.0, 0.0] for m in refs}
    for name, prm in model.named_parameters():
        if name.endswith("dc_weight"):
            continue
        g = prm.grad.detach().cpu().double()
        line = f"   {name:45s} |g| {float(g.norm()):.3e}"
        for m in refs:
            r = refs[m][1][name].double()
            line += f"  {m} {float((g - r).norm() / r.norm()):.2e}"
            tot[m][0] += float((g - r).norm() ** 2)
            tot[m][1] += float(r.norm() ** 2)
        print(line)
    print("   whole vector: " + "  ".join(f"{m} {(tot[m][0] / tot[m][1]) ** 0.5:.3e}" for m in refs), flush=True)
    flat = {m: torch.cat([refs[m][1][n].reshape(-1).double() for n, _ in model.named_parameters() if not n.endswith("dc_weight")]) for m in refs}
    ms = list(refs)
    print("   oracle vs oracle (this host's CPU): " + "  ".join(f"{a_}-{b_} {float((flat[a_] - flat[b_]).norm() / flat[b_].norm()):.3e}"
                                                              for i, a_ in enumerate(ms) for b_ in ms[i + 1:]), flush=True)
