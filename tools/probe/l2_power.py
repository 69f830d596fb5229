"""Is layer 2 bound by the chip's power budget?  The same launch (8 slices of 640 x 372, product library) on operands of different toggle activity: random
data, states that are exact fp16 values (the low terms of the split are zero), zero states, zero states and zero weights.  Identical instruction streams;
a time that follows the DATA is the clock the power management grants (MI355X_MICROARCH.md, DVFS give-back), not anything the kernel schedules."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
F, B, H, W = 64, 8, 640, 372


def timed(fn, n=int(os.environ.get("PROBE_N", "40"))):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


w2, wi2, wf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24
bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
z = torch.zeros_like
h1 = ops.cb8_from_nchw(r(B, F, H, W).relu())
hp = ops.cb8_from_nchw(r(B, F, H, W).relu())
o2, tp = torch.empty_like(h1), torch.empty(B, 18, H, W, device=dev)
cases = [("random states, random weights", h1, hp, (w2, wi2, wf), (bc, bi, hh)),
         ("states exact in fp16 (low terms zero)", h1.half().float(), hp, (w2, wi2, wf), (bc, bi, hh)),
         ("states and weights exact in fp16", h1.half().float(), hp, (w2.half().float(), wi2.half().float(), wf.half().float()), (bc, bi, hh)),
         ("zero states, random weights", z(h1), z(hp), (w2, wi2, wf), (bc, bi, hh)),
         ("zero states, zero weights", z(h1), z(hp), (z(w2), z(wi2), z(wf)), (z(bc), z(bi), z(hh)))]
for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    for name, x, hprev, ws, bs in cases:
        pk = ops.rim_layer2_f16_pack(*ws)
        xm = x.abs().max().reshape(1).contiguous()
        t = timed(lambda: ops.rim_layer2_f16_cb8(x, pk, bs[0], bs[1], bs[2], hprev, xm, taps=tp, out=o2, want_taps=True))
        print(f"{name:42s} {t / B:7.2f} us per slice", flush=True)
