"""The general-mask data-consistency gradient (three launches: mrx_pfa372_expand -> k_cols_dc_t4 -> k_pfa372_reduce; rim_utils.py:44-62 for a 2-D mask) at
15 x 640 x 372, B slices per launch (argv[1], default 4 = the 2-D-mask bench line's batch): three calls, for rocprofv3 --kernel-trace --stats (durations per
pass) and --pmc passes (SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES: is a pass parked on
memory or issuing vector instructions?).  tools/probe/llg2d_passes_summary.py turns the CSVs into profiles/r06_llg2d_pass_counters.md."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
B, C, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), 15, 640, 372
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
mask = (torch.rand(1, 1, H, W, 1, generator=g) < 0.1).to(dev)
y = y * mask
work = torch.empty_like(y)
torch.cuda.synchronize()
for _ in range(4):
    part, n = ops.llg(eta, y, S, mask, 1.0, False, "backward", work=work, parts=True)
torch.cuda.synchronize()
if os.environ.get("PROBE_TIME"):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.llg(eta, y, S, mask, 1.0, False, "backward", work=work, parts=True)
    e.record()
    torch.cuda.synchronize()
    print(f"B {B}: {1e3 * s.elapsed_time(e) / 20 / B:.2f} us per slice for the three launches (nparts {n})")
