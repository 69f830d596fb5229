"""Race screen for mrx_rim_layer2_cb8: repeated launches against the NCHW kernel's (bit-identical) results, mismatches located."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, F, H, W = 1, 64, int(os.environ.get("PROBE_H", "640")), int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x, hp = r(B, F, H, W).relu() * 1e-3, r(B, F, H, W).relu() * 1e-3
wc, wi = r(F, F, 3, 3) / 24 * 30, r(F, F, 1, 1) / 8
bc, bi, hh = r(F) * 0.1 * 0.03, r(F) * 0.1, r(1, F, 1, 1) * 0.5
wf = r(2, F, 3, 3) / 24
pk = ops.rim_layer2_f16_pack(wc, wi, wf)
xmax = x.abs().max().reshape(1).contiguous()
xc, hpc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
ref_h, ref_t = ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, xmax, want_taps=True)
ref_h0 = ops.rim_layer2_f16(x, pk, bc, bi, hh, None, xmax)
torch.cuda.synchronize()
for it in range(int(os.environ.get("PROBE_ITERS", "30"))):
    out = torch.full((B, 8, H, W, 8), float("nan"), device=dev)
    taps = torch.full((B, 9, H, W, 2), float("nan"), device=dev)
    ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, taps=taps, out=out, want_taps=True)
    h = ops.cb8_to_nchw(out)
    out0 = torch.full((B, 8, H, W, 8), float("nan"), device=dev)
    ops.rim_layer2_cb8(xc, pk, bc, bi, hh, None, xmax, out=out0)
    h0 = ops.cb8_to_nchw(out0)
    t = taps.permute(0, 1, 4, 2, 3).reshape(B, 18, H, W)
    out1 = torch.full((B, 8, H, W, 8), float("nan"), device=dev)
    ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, out=out1)
    h1 = ops.cb8_to_nchw(out1)
    out2 = torch.full((B, 8, H, W, 8), float("nan"), device=dev)
    ops.rim_layer2_cb8(xc, pk, bc, bi, hh, None, xmax, taps=taps, out=out2, want_taps=True)
    h2 = ops.cb8_to_nchw(out2)
    for name, got, ref in (("h", h, ref_h), ("h zero state", h0, ref_h0), ("taps", t, ref_t), ("h no taps", h1, ref_h), ("h zero state with taps", h2, ref_h0)):
        bad = ~(got == ref)
        if name == "taps":
            bad = (got - ref).abs() > 1e-5 * ref.abs().max()
        n = int(bad.sum())
        if n:
            idx = bad.nonzero()
            ys, xs, cs = idx[:, 2], idx[:, 3], idx[:, 1]
            print("iter %d %s: %d bad elements; rows %d..%d cols %d..%d channels %s; tiles (y//8, x//32): %s" % (
                it, name, n, int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()), sorted(set(cs.tolist()))[:16],
                sorted(set(zip((ys // 8).tolist(), (xs // 32).tolist())))[:12]), flush=True)
            print("     x%32:", sorted(set((xs % 32).tolist())), " got", got[bad][:6].tolist(), " ref", ref[bad][:6].tolist(), flush=True)
print("done")
