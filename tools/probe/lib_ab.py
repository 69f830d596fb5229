"""Which entry point of a source behaves differently between two builds of the library?  Runs the fp32 explicit training tape (seed 5, boosted
weights, 4 x 48 x 40: the configuration that fails with conv_bwd.hip built without packed-fp32 instructions) on library A with ONE function at a
time taken from library B, and prints the whole-gradient error against the fp32 oracle.
usage: python tools/probe/lib_ab.py <libB.so> <source.hip>"""
import ctypes
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from mridc_amd import _lib, synthetic, training
from mridc_amd.collections.reconstruction.models.cirim import CIRIM

libB_path, source = sys.argv[1], sys.argv[2]
A = _lib.lib()
B = ctypes.CDLL(libB_path)
for name, (args, res) in _lib._SIGNATURES.items():
    fn = getattr(B, name)
    fn.argtypes, fn.restype = args, res
src = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", source)).read()
names = sorted(set(re.findall(r'extern "C" [a-z0-9_]+ (mrx_[a-z0-9_]+)\(', src)))


class Hybrid:
    def __init__(self, swap):
        self.swap = set(swap)

    def __getattr__(self, n):
        return getattr(B if n in self.swap else A, n)


dev = torch.device("cuda:0")
cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)
torch.manual_seed(5)
model = CIRIM(cfg)
with torch.no_grad():
    for n_, p_ in model.named_parameters():
        if n_.endswith("bias"):
            p_.normal_(0, 0.05)
        if n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh"):
            p_.mul_(3.0)
state = {k: v.detach().clone() for k, v in model.state_dict().items()}
s = synthetic.make_slice(4, 48, 40, slice_idx=7)
ref_loss, ref = oracle.amp.cirim_loss_and_gradients(state, cfg, s, "fp32")
model = model.to(dev).train()
batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
pn = [n for n, _ in model.named_parameters() if not n.endswith("dc_weight")]
want = torch.cat([ref[n].reshape(-1).double() for n in pn])


def run(swap):
    _lib._lib = Hybrid(swap)
    for prm in model.parameters():
        prm.grad = None
    loss = training.cirim_forward_backward(model, batch, "f32")
    got = torch.cat([dict(model.named_parameters())[n].grad.detach().cpu().reshape(-1).double() for n in pn])
    _lib._lib = A
    return float((got - want).norm() / want.norm()), float(loss)


print("library A alone:", run(()))
print("library B alone:", run(_lib._SIGNATURES.keys()))
print(f"all {len(names)} entry points of {source} from B:", run(names))
for n in names:
    e, l_ = run((n,))
    print(f"   only {n:36s} from B: whole-gradient error {e:.3e}" + ("   <-- fixes it" if e < 1e-3 else ""), flush=True)
