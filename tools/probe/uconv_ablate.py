"""k_uconv at the E2EVN shapes under rocprofv3-free timing: many back-to-back launches between two events (launch overhead hidden by the queue)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
L = _lib.lib()
for (cin, cout, H, W) in [(14, 14, 640, 380), (28, 14, 640, 380), (28, 28, 320, 190), (56, 56, 160, 95), (18, 18, 640, 384)]:
    x, w = r(1, cin, H, W), r(cout, cin, 3, 3) / (cin * 9) ** 0.5
    nrm = torch.stack([r(1, cin) * 0.1, r(1, cin).abs() + 0.5], -1)
    y = torch.empty(1, cout, H, W, device=dev)
    norm = torch.empty(1, cout, 2, device=dev)
    work = torch.empty(int(L.mrx_unet_conv3x3_work_floats(1, cout, H, W)), device=dev)
    st = _lib.stream_ptr()
    def fn():
        L.mrx_unet_conv3x3(_lib.ptr(x), _lib.ptr(nrm), cin, None, None, 0, _lib.ptr(w), _lib.ptr(y), _lib.ptr(norm), _lib.ptr(work), 1, cout, H, W, 1e-5, 0.2, st)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"ablate {os.environ.get('MRX_UCONV_ABLATE', '0')}: {cin:3d}->{cout:3d} @{H}x{W}: {5 * s.elapsed_time(e):6.1f} us per (conv + finalize)")
