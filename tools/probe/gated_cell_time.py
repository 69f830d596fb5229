"""ConvGRUCell / ConvMGUCell 1x1 on 64 features at 640 x 372: split-bf16 kernel (default) against the fp32-MFMA kernel (MRIDC_AMD_ARITH=fp32)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, h = r(1, 64, 640, 372), r(1, 64, 640, 372)
for gates in (3, 2):
    wi, wh, bi = r(gates * 64, 64, 1, 1) / 8, r(gates * 64, 64, 1, 1) / 8, r(gates * 64) * 0.1
    pk = ops.gated_cell_pack(wi, wh, gates)
    fn = lambda: ops.gated_cell_1x1(x, h, pk, bi, gates)  # noqa: E731
    # float64 reference
    xd, hd = x.double(), h.double()
    ih = torch.nn.functional.conv2d(xd, wi.double(), bi.double())
    hh = torch.nn.functional.conv2d(hd, wh.double())
    if gates == 3:
        i_r, i_z, i_n = ih.chunk(3, 1); h_r, h_z, h_n = hh.chunk(3, 1)
        rg, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + rg * h_n)
        ref = n * (1 - z) + z * hd
    else:
        i_f, i_c = ih.chunk(2, 1); h_f, h_c = hh.chunk(2, 1)
        f = torch.sigmoid(i_f + h_f)
        c = torch.tanh(i_c + f * h_c)
        ref = c + f * (hd - c)
    out = fn()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100): fn()
    e.record(); torch.cuda.synchronize()
    print(f"gates {gates} fp32={os.environ.get('MRIDC_AMD_ARITH', 'f16x2')}: rel-L2 vs float64 {float((out.double() - ref).norm() / ref.norm()):.3e}, {10 * s.elapsed_time(e):.1f} us per launch")
