"""Time the general-mask (2-D) log-likelihood gradient at 15 x 640 x 372: column-tiled coil stack vs the row-major three-pass form,
pass by pass (HIP events over back-to-back launches)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import _lib, ops

dev = torch.device("cuda:0")
B, C, H, W = 1, int(os.environ.get("C", 15)), int(os.environ.get("H", 640)), 372
g = torch.Generator().manual_seed(0)
y = torch.randn(B, C, H, W, 2, generator=g).to(dev)
S = torch.randn(B, C, H, W, 2, generator=g).to(dev)
eta = torch.randn(B, H, W, 2, generator=g).to(dev)
mask = (torch.rand(1, 1, H, W, 1, generator=g) < 0.3).to(dev)
y = y * mask


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


L = _lib.lib()
sp = ops._sp372(S, False)
yt4 = ops._y_t4(y)
work = torch.empty_like(y)
wk = torch.empty(int(L.mrx_llg372_work_floats(B, C, H)), dtype=torch.float32, device=dev)
out = torch.empty(B, 4, H, W, device=dev)
m, kind, ms = _lib.mask_args(mask, B, C, H, W)
st = _lib.stream_ptr()
P = _lib.ptr
a = ops.llg(eta, y, S, mask, 1.0, False, "backward")
ops.LLG_T4 = False
b = ops.llg(eta, y, S, mask, 1.0, False, "backward")
print("tiled == row-major:", bool(torch.equal(a, b)))
print("row-major whole gradient: %.2f us" % timeit(lambda: ops.llg(eta, y, S, mask, 1.0, False, "backward", out=out, work=work)))
print("  expand      %.2f us" % timeit(lambda: L.mrx_pfa372_expand(P(eta), P(sp), P(work), None, None, None, 0, None, None, B, C, H, 0, 0, st)))
print("  cols + DC   %.2f us" % timeit(lambda: L.mrx_llg_cols_dc(P(work), P(y), P(m), kind, ms, B, C, H, W, 0, 0, st)))
print("  reduce+comb %.2f us" % timeit(lambda: L.mrx_pfa372_reduce(P(work), P(sp), P(eta), None, P(out), P(wk), B, C, H, 1.0, 0, 0, st)))
ops.LLG_T4 = True
print("column-tiled whole gradient: %.2f us" % timeit(lambda: ops.llg(eta, y, S, mask, 1.0, False, "backward", out=out, work=work)))
print("  expand      %.2f us" % timeit(lambda: L.mrx_pfa372_expand_t4(P(eta), P(sp), P(work), B, C, H, 0, 0, st)))
print("  cols + DC   %.2f us" % timeit(lambda: L.mrx_llg_cols_dc_t4(P(work), P(yt4), P(m), kind, ms, B, C, H, W, 0, 0, st)))
print("  reduce+comb %.2f us" % timeit(lambda: L.mrx_pfa372_reduce_t4(P(work), P(sp), P(eta), P(out), P(wk), None, B, C, H, 1.0, 0, 0, st)))
import ctypes
n = ctypes.c_int(0)
print("  reduce only %.2f us" % timeit(lambda: L.mrx_pfa372_reduce_t4(P(work), P(sp), None, None, P(wk), ctypes.byref(n), B, C, H, 1.0, 0, 0, st)))
