"""Phase trace (MRX_TRACE) and timing of the first RIM layer (k_rim_layer<5,1,4>) at 1 x 640 x 372 with the gradient partial sums as its input."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, 372
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
eta, part, hp = r(B, H, W, 2), r(3, B, H, W, 2), r(B, F, H, W)
pk1 = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8)
bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
fn = lambda: ops.rim_layer_indrnn_packed_llg(eta, part, 3, 1.0, pk1, F, 5, 1, bc, bi, hh, hp)  # noqa: E731
for _ in range(10):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100):
    fn()
e.record()
torch.cuda.synchronize()
print("layer 1: %.2f us per launch" % (10 * s.elapsed_time(e)))
os.environ["MRX_TRACE"] = "1"
os.environ["MRX_TRACE_DUMP"] = "1"
fn()
torch.cuda.synchronize()
