"""Time the bf16-operand convolutions against their fp32 counterparts at the headline shape."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = 1, 64, 640, 372
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = r(B, F, H, W), r(B, F, H, W)
x4 = r(B, 4, H, W)
w3, w1, w5 = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(F, 4, 5, 5) / 10
bias, hh = r(F), r(1, F, 1, 1)


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


print("3x3 d2 64->64 + ReLU : bf16 %.1f us | fp32 (Winograd) %.1f us" % (
    timeit(lambda: ops.conv2d_bf16(x, w3, bias, 2, ops.PAD_REPLICATE, ops.ACT_RELU)),
    timeit(lambda: ops.conv2d(x, w3, bias, 2, ops.PAD_REPLICATE, ops.ACT_RELU))))
print("1x1 IndRNN cell      : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv2d_bf16(x, w1, bias, 1, ops.PAD_ZERO, ops.ACT_RELU, hh=hh, h_prev=hp)),
    timeit(lambda: ops.indrnn_cell(x, w1, bias, hh, hp, 1))))
print("5x5 4->64 + ReLU     : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv2d_bf16(x4, w5, bias, 1, ops.PAD_REPLICATE, ops.ACT_RELU)),
    timeit(lambda: ops.conv2d(x4, w5, bias, 1, ops.PAD_REPLICATE, ops.ACT_RELU))))
print("dgrad 3x3 d2 64->64  : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv2d_bf16(x, w3, None, 2, ops.PAD_ZERO, transposed=True)),
    timeit(lambda: ops.conv2d(x, w3.flip(2, 3).transpose(0, 1).contiguous(), None, 2, ops.PAD_ZERO))))
print("dgrad 5x5 64->4      : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv2d_bf16(x, w5, None, 1, ops.PAD_ZERO, transposed=True)),
    timeit(lambda: ops.conv2d(x, w5.flip(2, 3).transpose(0, 1).contiguous(), None, 1, ops.PAD_ZERO))))
dy = r(B, F, H, W)
print("wgrad 3x3 d2 64->64  : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv_wgrad_bf16(x, dy, 3, 2, ops.PAD_REPLICATE), 30), timeit(lambda: ops.conv_wgrad(x, dy, 3, 2, ops.PAD_REPLICATE), 30)))
print("wgrad 1x1 64->64     : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv_wgrad_bf16(x, dy, 1, 1, ops.PAD_ZERO), 30), timeit(lambda: ops.conv_wgrad(x, dy, 1, 1, ops.PAD_ZERO), 30)))
dy2, dy64 = r(B, 2, H, W), dy
print("wgrad 3x3 64->2      : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv_wgrad_bf16(x, dy2, 3, 1, ops.PAD_REPLICATE), 30), timeit(lambda: ops.conv_wgrad(x, dy2, 3, 1, ops.PAD_REPLICATE), 30)))
print("wgrad 5x5 4->64      : bf16 %.1f us | fp32 %.1f us" % (
    timeit(lambda: ops.conv_wgrad_bf16(x4, dy64, 5, 1, ops.PAD_REPLICATE), 30), timeit(lambda: ops.conv_wgrad(x4, dy64, 5, 1, ops.PAD_REPLICATE), 30)))
