"""Probe: two slices reconstructed concurrently (one captured hipGraph per stream) vs back to back."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from mridc_amd import synthetic, _lib
from mridc_amd.collections.reconstruction.models.cirim import CIRIM
dev = torch.device('cuda:0')
cfg = dict(synthetic.CIRIM_BASELINE_CFG)
torch.manual_seed(0)
model = CIRIM(cfg).eval().to(dev)
C, H, W = 15, 640, 372
_lib.check(_lib.lib().mrx_fft_prepare(H, W), "prep")
def mk(i):
    s = synthetic.make_slice(C, H, W, slice_idx=i)
    return {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data = [mk(i) for i in range(NS)]
def step(d):
    with torch.no_grad():
        return next(model(d["y"], d["sensitivity_maps"], d["mask"], None, d["target"]))
for d in data:
    step(d)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(NS)]
graphs = []
for d, st in zip(data, streams):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        step(d)
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        step(d)
    graphs.append(g)
torch.cuda.synchronize()
def run_seq(n):
    t0 = time.perf_counter()
    for _ in range(n):
        for g in graphs:
            g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (NS * n)
def run_par(n):
    t0 = time.perf_counter()
    for _ in range(n):
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (NS * n)
run_seq(2); run_par(2)
print(f"sequential: {1/run_seq(5):.2f} slices/s   {NS} streams: {1/run_par(5):.2f} slices/s")
