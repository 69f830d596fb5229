"""mrx_fft_cols (IFFT along H of a 8 x 15 x 640 x 372 coil stack: the once-per-slice transform of the hybrid-space gradient) by HIP events, and its result against torch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
y = torch.randn(8, 15, 640, 372, 2, generator=g).to(dev)
out = ops.llg_prepare(y, True, "ortho")
ref = torch.view_as_real(torch.fft.fftshift(torch.fft.ifft(torch.fft.ifftshift(torch.view_as_complex(y[:1].double().cpu()), dim=-2), dim=-2, norm="ortho"), dim=-2))
err = float((out[:1].double().cpu() - ref).norm() / ref.norm())
for _ in range(3):
    ops.llg_prepare(y, True, "ortho")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.llg_prepare(y, True, "ortho")
e1.record()
torch.cuda.synchronize()
lib = os.path.basename(os.path.dirname(os.environ.get("MRIDC_AMD_LIB", "mridc_amd/lib/x")))
print(f"{lib:10s} mrx_fft_cols 8 x 15 x 640 x 372: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us   rel-L2 vs float64 {err:.2e}")
