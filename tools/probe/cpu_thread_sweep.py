"""One-off: the CPU oracle (one CIRIM cascade of 8 time-steps at 15 x 640 x 372, after a warm-up cascade) at 16 / 32 / 64 / 128 / 256 torch threads on
the GPU box's host.  BASELINE.md section 3 says torch.set_num_threads(os.cpu_count()); bench.py's cpu_baseline uses the fastest setting of this
sweep (bench.ORACLE_THREADS_DEFAULT) -- the output is committed as profiles/r05_cpu_thread_sweep.txt."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from mridc_amd import synthetic  # noqa: E402
from mridc_amd.collections.reconstruction.models.cirim import CIRIM  # noqa: E402

cfg = dict(synthetic.CIRIM_BASELINE_CFG)
torch.manual_seed(0)
state = {k: v.detach().clone() for k, v in CIRIM(cfg).state_dict().items()}
s = synthetic.make_slice(15, 640, 372, slice_idx=0)
box = os.cpu_count() or 1
print(f"host: {box} hardware threads")
best = None
for n in [t for t in (8, 16, 32, 64, 128, 256) if t <= box] or [box]:
    torch.set_num_threads(n)
    with torch.no_grad():
        oracle.models.cirim_forward(state, dict(cfg, num_cascades=1), s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])      # warm-up
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            oracle.models.cirim_forward(state, dict(cfg, num_cascades=1), s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
            ts.append(time.perf_counter() - t0)
    sec = min(ts)
    print(f"threads {n:4d}: {sec:7.2f} s per cascade (8 RIM steps) = {1.0 / (8 * sec):.4f} slices/s extrapolated to 8 cascades   (runs: {', '.join(f'{t:.2f}' for t in ts)})", flush=True)
    if best is None or sec < best[1]:
        best = (n, sec)
print(f"fastest: {best[0]} threads")
