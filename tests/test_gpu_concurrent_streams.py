"""Kernels of different slices sharing the GPU must not influence each other: whole-model hipGraphs replayed CONCURRENTLY on separate streams return
bit for bit what the same graphs return one at a time.

Round-3 finding behind this file (tools/probe/mfma_pk_interference.py): on MI355X a wave executing packed-fp32 vector instructions (v_pk_fma_f32 ...:
the FFT kernels were built on them) returns wrong results while a wave of another kernel on the same SIMD issues XDL MFMAs (the U-Net and
few-channel convolutions of another slice: several small workgroups per CU) -- 1e-3 .. 5e-2 errors in E2EVN on two streams, invisible to every
single-stream test.  The library is built without packed-fp32 instructions (mridc_amd/_build.py: NO_PACKED_FP32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _concurrent_vs_serial(fns, reps=4):
    """Captures every fn on its own stream, replays all graphs concurrently `reps` times; returns the number of replays whose output differs from
    the eager result of the same fn."""
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
    bad = 0
    for _ in range(reps):
        for g_, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g_.replay()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, r)) for o, r in zip(outs, refs))
    return bad


def test_fft_kernels_next_to_matrix_core_kernels_of_another_stream(dev):
    """The minimal form: row / column transforms and the coil reduction (victims) next to a chain of U-Net convolutions and next to the
    few-channel convolution (aggressors), 15 x 640 x 372."""
    from mridc_amd import ops, synthetic
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    d = {k: (torch.cat([v] * 2, 0) if k != "mask" else v).to(dev) for k, v in synthetic.make_slice(15, 640, 372, slice_idx=0).items()}
    yh = ops.llg_prepare(d["y"], False, "backward", [-2, -1])
    red = ops.sens_reduce(yh, d["sensitivity_maps"], False, "backward", [-2, -1], hybrid=True)
    one = torch.ones(1, device=dev)
    a14 = r(2, 14, 640, 384)
    n14 = torch.stack([a14.mean((2, 3)), 1 / torch.sqrt(a14.var((2, 3), unbiased=False) + 1e-5)], -1)
    w14 = r(14, 14, 3, 3) / 11
    x8, w8, b8 = r(2, 8, 640, 372), r(128, 8, 5, 5) / 14, r(128) * 0.1

    def rep(fn, n):
        def f():
            o = None
            for _ in range(n):
                o = fn()
            return o
        return f

    def unet_chain():
        o = (a14, n14)
        for _ in range(10):
            o = ops.unet_conv3x3(o, None, w14)
        return o[0]

    prep = rep(lambda: ops.llg_prepare(d["y"], False, "backward", [-2, -1]), 6)
    reduce_ = rep(lambda: ops.sens_reduce(yh, d["sensitivity_maps"], False, "backward", [-2, -1], hybrid=True), 6)
    expand = rep(lambda: ops.sens_expand_dc_hybrid(red.unsqueeze(1), d["sensitivity_maps"], yh, yh, d["mask"], one, False, "backward", reduce=True)[0], 6)
    sbs = rep(lambda: ops.conv_sbs(x8, w8, b8, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0), 10)
    for aggressor in (unet_chain, sbs):
        assert _concurrent_vs_serial([prep, aggressor, reduce_, expand]) == 0


@pytest.mark.parametrize("which,streams", [("e2evn", 2), ("e2evn", 3), ("cirim", 2)])
def test_whole_models_on_concurrent_streams_are_bit_identical_to_serial(dev, which, streams):
    from mridc_amd import synthetic
    torch.manual_seed(0)
    if which == "cirim":
        from mridc_amd.collections.reconstruction.models.cirim import CIRIM
        model = CIRIM(dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)).eval().to(dev)

        def run(d):
            return next(model(d["y"], d["sensitivity_maps"], d["mask"], None, d["target"]))[-1][-1]
    else:
        from mridc_amd.collections.reconstruction.models.vn import VarNet
        common = dict(fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE",
                      use_sens_net=False)
        model = VarNet(dict(synthetic.E2EVN_BASELINE_CFG, num_cascades=3, **common)).eval().to(dev)

        def run(d):
            return torch.view_as_real(model(d["y"], d["sensitivity_maps"], d["mask"], None, d["target"]))
    datas = [{k: v.to(dev) for k, v in synthetic.make_slice(15, 640, 372, slice_idx=i).items()} for i in range(streams)]
    assert _concurrent_vs_serial([(lambda d=d: run(d)) for d in datas], reps=3) == 0


def test_qcirim_on_concurrent_streams_is_bit_identical_to_serial(dev):
    """qCIRIM (few-channel convolution, 128-channel layers, the signal model's pointwise kernels, 256 x 256 transforms) on two streams."""
    import bench
    from mridc_amd.collections.quantitative.models.qcirim import qCIRIM
    torch.manual_seed(0)
    model = qCIRIM(bench.QCIRIM_CFG).eval().to(dev)
    E, C, H, W = 4, 32, 256, 256
    TEs = [3.0, 11.5, 20.0, 28.5]
    datas = []
    for i in range(2):
        g = torch.Generator().manual_seed(100 + i)
        maps = [torch.rand(1, H, W, generator=g) * s for s in (0.3, 1.0, 0.1, 0.5)]
        S = torch.randn(1, C, H, W, 2, generator=g) / C ** 0.5
        mask = (torch.rand(1, 1, 1, 1, W, 1, generator=g) < 0.3)
        y = torch.randn(1, E, C, H, W, 2, generator=g) * mask
        datas.append([t.to(dev) for t in maps + [y, S, mask]])

    def run(d):
        out = next(model(d[0], d[1], d[2], d[3], TEs, d[4], d[5], None, d[6]))
        return torch.stack([out[1 + m][-1][-1] for m in range(4)])

    assert _concurrent_vs_serial([(lambda d=d: run(d)) for d in datas], reps=6) == 0
