"""Time the NormUnet building blocks at the E2EVN (14, 2) shapes: conv3x3 (+ fused InstanceNorm statistics), the apply pass, the transpose conv."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


for (cin, cout, H, W) in [(2, 14, 640, 380), (14, 14, 640, 380), (28, 14, 640, 380), (14, 28, 320, 190), (28, 28, 320, 190), (56, 28, 320, 190),
                          (28, 56, 160, 95), (56, 56, 160, 95)]:
    x, w = r(1, cin, H, W), r(cout, cin, 3, 3) / (cin * 9) ** 0.5
    t_all = timeit(lambda: ops.conv_instance_norm_act(x, w, 1e-5, ops.ACT_LEAKY, 0.2))
    t_conv = timeit(lambda: ops.conv2d(x, w, None, 1, ops.PAD_ZERO))
    nrm = torch.stack([r(1, cin) * 0.1, r(1, cin).abs() + 0.5], -1)
    t_fused = timeit(lambda: ops.unet_conv3x3((x, nrm), None, w))
    mb = 4 * H * W * (cin + cout) / 1e6
    gf = 2 * cin * cout * 9 * H * W / 1e9
    print(f"conv3x3 {cin:3d}->{cout:3d} @{H}x{W}: conv+IN+act {t_all:6.1f} us | conv alone {t_conv:6.1f} us | lazy in -> (raw, norm) out {t_fused:6.1f} us | {mb:6.1f} MB -> {mb / 5e3 * 1e3:5.1f} us at 5 TB/s, "
          f"{gf:5.2f} GFLOP -> {gf / 157.3 * 1e3:5.1f} us at the fp32 MFMA peak")
for (cin, cout, H, W) in [(56, 28, 160, 95), (28, 14, 320, 190)]:
    x, w = r(1, cin, H, W), r(cin, cout, 2, 2) / (cin * 4) ** 0.5
    print(f"convT2x2 {cin}->{cout} @{H}x{W}: {timeit(lambda: ops.conv_transpose2x2(x, w)):6.1f} us; + IN + act "
          f"{timeit(lambda: ops.instance_norm_act(ops.conv_transpose2x2(x, w), 1e-5, ops.ACT_LEAKY, 0.2)):6.1f} us")
