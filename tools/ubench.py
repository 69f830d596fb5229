#!/usr/bin/env python3
"""Micro-benchmarks of individual HIP entry points at the headline size (run on the GPU box).

    python tools/ubench.py [layer2|layer1|final|llg|all]
"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mridc_amd import ops  # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ops_to_dev(x, dev):
    from mridc_amd.collections.common.parts import utils as U
    return U.to_tensor(x).to(dev)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda:0")
    B, C, H, W, F = 1, 15, 640, 372, 64
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    if what in ("layer2", "all"):
        x, hp = r(B, F, H, W), r(B, F, H, W)
        packed = ops.rim_layer_pack(r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8)
        bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
        out = torch.empty_like(hp)
        t = timeit(lambda: ops.rim_layer_indrnn_packed(x, packed, F, 3, 2, bc, bi, hh, hp, out=out))
        fl = 2.0 * (F * F * 9 + F * F) * H * W * B
        print(f"layer2 (3x3 d2 64->64 + ih): {t:.1f} us  {fl / t / 1e6:.1f} TFLOP/s  ablate={os.environ.get('MRX_ABLATE', '0')}")
    if what in ("wino", "layer2", "all"):
        x, hp = r(B, F, H, W), r(B, F, H, W)
        wc, wi = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8
        packed = ops.rim_layer_wino_pack(wc, wi)
        bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
        out = torch.empty_like(hp)
        t = timeit(lambda: ops.rim_layer_indrnn_wino(x, packed, F, bc, bi, hh, hp, out=out))
        fl = 2.0 * (F * F * 9 + F * F) * H * W * B
        direct = ops.rim_layer_indrnn_packed(x, ops.rim_layer_pack(wc, wi), F, 3, 2, bc, bi, hh, hp)
        err = ((out - direct).norm() / direct.norm()).item()
        print(f"layer2 winograd: {t:.1f} us  {fl / t / 1e6:.1f} direct-equivalent TFLOP/s  rel-L2 vs direct {err:.2e}")
    if what in ("layer1", "all"):
        x, hp = r(B, 4, H, W), r(B, F, H, W)
        packed = ops.rim_layer_pack(r(F, 4, 5, 5) / 10, r(F, F, 1, 1) / 8)
        bc, bi, hh = r(F), r(F), r(1, F, 1, 1)
        out = torch.empty_like(hp)
        t = timeit(lambda: ops.rim_layer_indrnn_packed(x, packed, F, 5, 1, bc, bi, hh, hp, out=out))
        fl = 2.0 * (4 * F * 25 + F * F) * H * W * B
        print(f"layer1 (5x5 4->64 + ih): {t:.1f} us  {fl / t / 1e6:.1f} TFLOP/s  ablate={os.environ.get('MRX_ABLATE', '0')}")
    if what in ("final", "all"):
        h, eta, w = r(B, F, H, W), r(B, H, W, 2), r(2, F, 3, 3) / 24
        t = timeit(lambda: ops.rim_final(h, w, None, 3, 1, eta))
        print(f"final (3x3 64->2 + eta): {t:.1f} us")
    if what in ("prep", "all"):
        # N1: per-sample preprocessing (target, masking, max-normalisation) on the device
        import time
        import numpy as np
        from mridc_amd.collections.reconstruction.data import subsample
        from mridc_amd.collections.reconstruction.parts.transforms import MRIDataTransforms
        rng = np.random.default_rng(5)
        k = (rng.standard_normal((C, H, W)) + 1j * rng.standard_normal((C, H, W))).astype(np.complex64)
        S_ = (rng.standard_normal((C, H, W)) + 1j * rng.standard_normal((C, H, W))).astype(np.complex64)
        kw = dict(normalize_inputs=True, max_norm=True, fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1)
        tr = MRIDataTransforms(mask_func=[subsample.RandomMaskFunc([0.08], [4])], **kw)
        kd, Sd = ops_to_dev(k, dev), ops_to_dev(S_, dev)
        t = timeit(lambda: tr(kd, Sd, None, np.array([]), np.array([]), {}, "file_7.h5", 0), n=10)
        t0 = time.perf_counter()
        tr(k, S_, None, np.array([]), np.array([]), {}, "file_7.h5", 0)
        torch.cuda.synchronize()
        t_h2d = (time.perf_counter() - t0) * 1e6
        print(f"preprocessing (15 x 640 x 372, SENSE target + random-1-D mask + max-normalisation): device {t:.0f} us per slice with the "
              f"inputs resident, {t_h2d:.0f} us from host NumPy arrays")
    if what in ("llg", "all"):
        eta, y, S = r(B, H, W, 2), r(B, C, H, W, 2), r(B, C, H, W, 2)
        mask = (torch.rand(1, 1, 1, W, 1) < 0.3).to(dev)
        work, out = torch.empty_like(y), torch.empty(B, 4, H, W, device=dev)
        t = timeit(lambda: ops.llg(eta, y, S, mask, 1.0, False, "backward", out=out, work=work))
        print(f"llg: {t:.1f} us  {(25 + 16 * C) * H * W * B / t / 1e3:.1f} GB/s (algorithmic)")
        yt = ops.llg_prepare(y, False, "backward")
        t = timeit(lambda: ops.llg_hinv(eta, yt, S, mask, 1.0, False, "backward", out=out))
        print(f"llg_hinv: {t:.1f} us  {(25 + 16 * C) * H * W * B / t / 1e3:.1f} GB/s (algorithmic)")
        t = timeit(lambda: ops.sens_expand(eta, S, False, "backward"))
        print(f"sens_expand: {t:.1f} us")


if __name__ == "__main__":
    main()
