"""Round-3 finding: kernels built on packed-fp32 vector ops (the FFT kernels) return WRONG results when their waves share a SIMD with waves of
another stream's kernel that issues XDL MFMAs (k_uconv_h, k_conv_sbs: several small workgroups per CU), and bit-exact ones otherwise.
Victims V (each a hipGraph of one op repeated), aggressor A, all replayed concurrently on separate streams vs one at a time.
VICTIM = prep | reduce | expand | llg372 | llg2d | fft2 | ifft2, AGGR = uconv | sbs | convT | pool."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops, synthetic
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
d = {k: (torch.cat([v] * 2, 0) if k != "mask" else v).to(dev) for k, v in synthetic.make_slice(15, 640, 372, slice_idx=0).items()}
YH = ops.llg_prepare(d["y"], False, "backward", [-2, -1])
RED = ops.sens_reduce(YH, d["sensitivity_maps"], False, "backward", [-2, -1], hybrid=True)
one = torch.ones(1, device=dev)
A14 = r(2, 14, 640, 384)
nA = torch.stack([A14.mean((2, 3)), 1 / torch.sqrt(A14.var((2, 3), unbiased=False) + 1e-5)], -1)
W14 = r(14, 14, 3, 3) / 11
X8, W8, B8 = r(2, 8, 640, 372), r(128, 8, 5, 5) / 14, r(128) * 0.1
A56 = r(2, 56, 160, 96)
nA56 = torch.stack([A56.mean((2, 3)), 1 / torch.sqrt(A56.var((2, 3), unbiased=False) + 1e-5)], -1)
WT = r(56, 28, 2, 2) / 15


def rep(fn, n):
    def f():
        o = None
        for _ in range(n):
            o = fn()
        return o
    return f


victims = dict(prep=rep(lambda: ops.llg_prepare(d["y"], False, "backward", [-2, -1]), 6),
               reduce=rep(lambda: ops.sens_reduce(YH, d["sensitivity_maps"], False, "backward", [-2, -1], hybrid=True), 6),
               expand=rep(lambda: ops.sens_expand_dc_hybrid(RED.unsqueeze(1), d["sensitivity_maps"], YH, YH, d["mask"], one, False, "backward", reduce=True)[0], 6))


from mridc_amd.collections.common.parts import fft as mfft
eta0 = r(2, 640, 372, 2)
m1 = ops.row_invariant_view(d["mask"])
op372 = ops.llg372_prepare(YH, d["sensitivity_maps"], m1, False)
mask2d = (torch.rand(1, 1, 640, 372, 1, generator=g) < 0.3).to(dev)
victims.update(llg372=rep(lambda: ops.llg372(eta0, op372, 1.0, "backward"), 6),
               llg2d=rep(lambda: ops.llg(eta0, d["y"], d["sensitivity_maps"], mask2d, 1.0, False, "backward"), 4),
               fft2=rep(lambda: mfft.fft2(d["y"], centered=True, normalization="ortho", spatial_dims=[-2, -1]), 4),
               ifft2=rep(lambda: mfft.ifft2(d["y"], centered=False, normalization="backward", spatial_dims=[-2, -1]), 4))


def chain():
    o = (A14, nA)
    for _ in range(10):
        o = ops.unet_conv3x3(o, None, W14)
    return o[0]


aggr = dict(uconv=chain, sbs=rep(lambda: ops.conv_sbs(X8, W8, B8, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0), 10),
            convT=rep(lambda: ops.unet_conv_transpose2x2((A56, nA56), WT)[0], 10), pool=rep(lambda: ops.unet_avg_pool2x2((A14, nA)), 10))
fns = [victims[os.environ.get("VICTIM", "prep")], aggr[os.environ.get("AGGR", "uconv")], victims[os.environ.get("VICTIM", "prep")]]
with torch.no_grad():
    refs = [f().clone() for f in fns]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in fns]
    graphs, outs = [], []
    for f, st in zip(fns, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            f()
        torch.cuda.current_stream().wait_stream(st)
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=st, capture_error_mode="thread_local"):
            outs.append(f())
        graphs.append(g_)
for mode in ("one at a time", "concurrent"):
    worst = [0.0] * 3
    for it in range(8):
        for g_, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g_.replay()
            if mode == "one at a time":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for i, (o, rf) in enumerate(zip(outs, refs)):
            worst[i] = max(worst[i], float((o.double() - rf.double()).norm() / rf.double().norm()))
    print(os.environ.get("VICTIM", "prep"), "x", os.environ.get("AGGR", "uconv"), mode, "(victim, aggressor, victim) worst rel-L2 vs own eager result:", worst, flush=True)
