"""What a HIP-event pair adds to a bracketed launch: empty pairs, and pairs around a kernel of known (rocprofv3) duration."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(1, 64, 640, 372, device=dev)
torch.cuda.synchronize()


def pairs(fn, n=200):
    ev = []
    torch.cuda._sleep(int(2e7))
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    t = sorted(1e3 * s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2], t[0]


print("empty pair: median %.2f us, min %.2f us" % pairs(lambda: None))
tiny = torch.zeros(64, device=dev)
print("tiny kernel (max_abs of 64 floats = 2 launches): median %.2f us, min %.2f us" % pairs(lambda: ops.max_abs(tiny)))
