"""Second RIM layer on channel-blocked states (mrx_rim_layer2_cb8) against the NCHW two-term fp16 kernel and float64: error and time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from mridc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, F, H, W = int(os.environ.get("PROBE_B", "1")), 64, int(os.environ.get("PROBE_H", "640")), int(os.environ.get("PROBE_W", "372"))
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timed(fn, n=100):
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


for xs, ws in ((1.0, 1.0), (1e-3, 30.0)):
    x, hp = r(B, F, H, W).relu() * xs, r(B, F, H, W).relu() * xs
    wc, wi = r(F, F, 3, 3) / 24 * ws, r(F, F, 1, 1) / 8
    bc, bi, hh = r(F) * 0.1 * xs * ws, r(F) * 0.1, r(1, F, 1, 1) * 0.5
    wf, bf, eta = r(2, F, 3, 3) / 24, r(2) * 0.1, r(B, H, W, 2)
    pk = ops.rim_layer2_f16_pack(wc, wi, wf)
    xmax = x.abs().max().reshape(1).contiguous()
    ref = Fn.relu(Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double(), dilation=2))
    ref = Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())
    ref_eta = eta.double() + (Fn.conv2d(Fn.pad(ref, (1, 1, 1, 1), mode="replicate"), wf.double()) + bf.double().view(1, 2, 1, 1)).permute(0, 2, 3, 1)
    xc, hpc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
    assert torch.equal(ops.cb8_to_nchw(xc), x)
    old_h, old_t = ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, xmax, want_taps=True)
    old_eta = ops.rim_final_gather(old_t, bf, eta)
    new_h, new_t = ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, want_taps=True)
    new_eta = ops.rim_final_gather9(new_t, bf, eta)
    new_h = ops.cb8_to_nchw(new_h)
    e = lambda a, b: ((a.double() - b).norm() / b.norm()).item()  # noqa: E731
    print("scales x %g w %g: h: nchw kernel %.3e, cb8 kernel %.3e; eta: %.3e / %.3e; cb8 vs nchw h %.3e" % (
        xs, ws, e(old_h, ref), e(new_h, ref), e(old_eta, ref_eta), e(new_eta, ref_eta), e(new_h, old_h.double())), flush=True)
    h0 = ops.cb8_to_nchw(ops.rim_layer2_cb8(xc, pk, bc, bi, hh, None, xmax))
    ref0 = Fn.relu(Fn.conv2d(Fn.relu(Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double(), dilation=2)), wi.double(), bi.double()))
    print("   zero state, no taps: %.3e" % e(h0, ref0), flush=True)
    taps_o, taps_n = torch.empty_like(old_t), torch.empty_like(new_t)
    out_o, out_n = torch.empty_like(old_h), torch.empty_like(xc)
    print("   time: nchw %.2f us, cb8 %.2f us (in place %.2f us), gather %.2f / %.2f us" % (
        timed(lambda: ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, xmax, taps=taps_o, out=out_o, want_taps=True)),
        timed(lambda: ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, taps=taps_n, out=out_n, want_taps=True)),
        timed(lambda: ops.rim_layer2_cb8(xc, pk, bc, bi, hh, out_n, xmax, taps=taps_n, out=out_n, want_taps=True)),
        timed(lambda: ops.rim_final_gather(old_t, bf, eta)), timed(lambda: ops.rim_final_gather9(new_t, bf, eta))), flush=True)
os.environ["MRX_L2C8_TRACE"] = "1"
ops.rim_layer2_cb8(xc, pk, bc, bi, hh, hpc, xmax, taps=taps_n, out=out_n, want_taps=True)
torch.cuda.synchronize()
