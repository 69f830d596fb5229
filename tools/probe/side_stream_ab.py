"""Training step with the weight gradients on a second stream: does it survive a process that already owns a dozen streams (bench.py's default line)?
HIP maps streams onto a few hardware queues; a side stream that lands on the main stream's queue serialises instead of overlapping."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import autograd as ag
from mridc_amd import synthetic, training
from mridc_amd.collections.reconstruction.models.cirim import CIRIM
dev = torch.device("cuda:0")
crowd = int(sys.argv[1]) if len(sys.argv) > 1 else 12
others = [torch.cuda.Stream() for _ in range(crowd)]
for st in others:
    with torch.cuda.stream(st):
        torch.zeros(1024, device=dev).add_(1)
torch.cuda.synchronize()
cfg = dict(synthetic.CIRIM_BASELINE_CFG)
torch.manual_seed(0)
model = CIRIM(cfg).to(dev)
flat = training.FlatParameters(model)
opt = training.AdamFlat(flat, lr=1e-3, betas=(0.9, 0.98))
s = synthetic.make_slice(15, 640, 372, slice_idx=0)
batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
ag.set_precision("bf16")


def run(label):
    training.training_step(model, flat, opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        training.training_step(model, flat, opt, batch)
    torch.cuda.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per step", flush=True)


training.TL_SIDE_STREAM = False
run(f"{crowd} other streams, no side stream")
training.TL_SIDE_STREAM = True
training._SIDE.clear()
run("side stream, default priority")
training._SIDE.clear()
training._SIDE[str(dev)] = torch.cuda.Stream(device=dev, priority=-1)
run("side stream, high priority")
training._SIDE.clear()
training._SIDE[str(dev)] = others[0]
run("side stream = the first stream the process created")
