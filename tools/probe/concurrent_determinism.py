"""Bitwise determinism of whole-model hipGraphs replayed concurrently on several streams against the same graphs replayed one at a time
(kernels of different slices sharing CUs must not influence each other).  MODEL=cirim | e2evn, NS = number of streams."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
which, NS, B = os.environ.get("MODEL", "cirim"), int(os.environ.get("NS", "2")), int(os.environ.get("PB", "1"))
if which == "cirim":
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    model = CIRIM(dict(synthetic.CIRIM_BASELINE_CFG)).eval().to(dev)

    def step(d):
        with torch.no_grad():
            return next(model(d["y"], d["sensitivity_maps"], d["mask"], None, d["target"]))[-1][-1]
else:
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    common = dict(fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE", use_sens_net=False)
    model = VarNet(dict(synthetic.E2EVN_BASELINE_CFG, **common)).eval().to(dev)

    def step(d):
        with torch.no_grad():
            return torch.view_as_real(model(d["y"], d["sensitivity_maps"], d["mask"], None, d["target"]))
datas = []
for i in range(NS):
    d = synthetic.make_slice(15, 640, 372, slice_idx=i)
    datas.append({k: (torch.cat([v] * B, 0) if k != "mask" else v).to(dev) for k, v in d.items()})
refs = [step(d).clone() for d in datas]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in datas]
graphs, outs = [], []
for d, st in zip(datas, streams):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        step(d)
    torch.cuda.current_stream().wait_stream(st)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_, stream=st, capture_error_mode="thread_local"):
        outs.append(step(d))
    graphs.append(g_)


def rel(a, b):
    if a.is_complex():
        a, b = torch.view_as_real(a), torch.view_as_real(b)
    return float((a.double() - b.double()).norm() / b.double().norm())


for mode in ("one at a time", "concurrent"):
    worst, nbad = [0.0] * NS, [0] * NS
    for it in range(int(os.environ.get("REPS", "6"))):
        for g_, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g_.replay()
            if mode == "one at a time":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, refs)):
            worst[i] = max(worst[i], rel(o, r))
            nbad[i] += int(not torch.equal(o, r))
    print(which, mode, "replays differing from the eager result:", nbad, "worst rel-L2:", worst, flush=True)
