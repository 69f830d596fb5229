"""Layer-2 weight gradient of the training tape (3x3 d2 64 -> 64, dy a pair tensor, x channel-blocked) at 1 x 640 x 372: one launch in each `accumulate`
mode (0/1 = with the 256-slot reduction, 2 = slots stored, 3 = slots added) and the reduction alone, HIP events, isolated."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mridc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 1, 640, 372
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 8, H, W, 8, generator=g).to(dev)
dy = ops.f32_to_pairs(torch.randn(B, 64, H, W, generator=g).bfloat16().float().to(dev))
out = torch.zeros(64, 64, 3, 3, device=dev)
L = _lib.lib()
nwork = int(L.mrx_conv_wgrad_bf16_any_work_floats(B, 64, 64, H, W, 3))
work = torch.zeros(nwork, device=dev)
total = 64 * 64 * 9


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def call(mode):
    _lib.check(L.mrx_conv_wgrad_bf16_pairs(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(out), _lib.ptr(work), B, 64, H, W, 3, 2, ops.PAD_REPLICATE, mode, 1, _lib.stream_ptr()), "pairs")


def red():
    _lib.check(L.mrx_wgrad_parts_reduce(_lib.ptr(work), nwork // total, total, _lib.ptr(out), 1, _lib.stream_ptr()), "reduce")


print(f"slots {nwork // total}  launch+reduce {t(lambda: call(1)):.1f} us   store only {t(lambda: call(2)):.1f}   add {t(lambda: call(3)):.1f}   reduce alone {t(red):.1f}")
