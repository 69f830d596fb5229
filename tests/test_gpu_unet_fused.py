"""GPU parity of the U-Net kernels that keep Conv -> InstanceNorm2d -> LeakyReLU(0.2) outputs as (raw, per-plane statistics) pairs and
normalise on load (csrc/unet_fused.hip; reference unet_base/unet_block.py:139-308): every operator against float64 torch, the whole
Unet / NormUnet against the conv + apply formulation (MRIDC_AMD_UNET_FUSED=0) and -- through the golden tests of test_gpu_models.py,
which now run this path by default -- against the reference.  Tolerances: operators 2e-6 of the output norm (fp32 MFMA = exact fp32 FMA
chains), whole networks 2e-5 (ten normalised layers deep)."""
import pytest
import torch
import torch.nn.functional as Fn

from tests._util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _norm_act(x, eps=1e-5, slope=0.2):
    return Fn.leaky_relu(Fn.instance_norm(x, eps=eps), slope)


def _lazy_ref(raw64, norm):
    """leaky((raw - mean) / std) from a kernel's own (mean, 1/std), in float64."""
    return Fn.leaky_relu((raw64 - norm[..., 0, None, None].double()) * norm[..., 1, None, None].double(), 0.2)


@pytest.mark.parametrize("shape", [(1, 2, 0, 14, 640, 380), (1, 14, 0, 14, 640, 380), (1, 14, 14, 14, 640, 380), (2, 14, 0, 28, 320, 190),
                                   (1, 28, 28, 28, 160, 95), (1, 56, 0, 56, 33, 47), (1, 18, 18, 18, 72, 40), (3, 5, 3, 36, 9, 7),
                                   (1, 144, 144, 144, 40, 24), (1, 3, 0, 70, 8, 32), (1, 1, 0, 1, 1, 2),
                                   # 16-row work items (one output-channel block and >= 2048 of them): E2EVN's batch of 8; a last item with its second 8-row half
                                   # below the image (200 = 12 x 16 + 8) and one with three rows of it inside (203)
                                   (8, 14, 14, 14, 640, 372), (8, 14, 0, 14, 200, 700), (8, 6, 3, 9, 203, 690)])
def test_unet_conv3x3_sources_and_statistics(shape, dev):
    """mrx_unet_conv3x3: plain and lazy sources, one and two of them (the skip concatenation read in place), every cout-block count, ragged
    tiles, channel counts that are not multiples of the 8-channel step; raw output and (mean, 1/std) against float64."""
    from mridc_amd import ops
    B, Ca, Cb, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    a_raw, b_raw = r(B, Ca, H, W) * 2 + 0.5, (r(B, Cb, H, W) - 0.3 if Cb else None)
    na = torch.stack([r(B, Ca) * 0.5, r(B, Ca).abs() + 0.5], -1)
    nb = torch.stack([r(B, Cb) * 0.5, r(B, Cb).abs() + 0.5], -1) if Cb else None
    w = r(Cout, Ca + Cb, 3, 3) / (9 * (Ca + Cb)) ** 0.5

    def true_stats(t):            # what the producing kernels hand over: (mean, 1 / sqrt(var + eps)) of every plane
        return torch.stack([t.mean((2, 3)), 1.0 / torch.sqrt(t.var((2, 3), unbiased=False) + 1e-5)], -1)

    keep = ops.UNET_F16
    try:
        # the fp32-input MFMA kernel takes ANY (mean, 1/std) pair; the two-term fp16 kernel (the default) bounds a lazy source by sqrt(n), which
        # holds for the statistics of the planes themselves -- the only pairs this library produces
        for f16 in (False, True):
            ops.UNET_F16 = f16
            if f16 and H * W > 1:
                na, nb = true_stats(a_raw), (true_stats(b_raw) if Cb else None)
            for lazy_a, lazy_b in ((False, False), (True, True), (True, False)):
                if Cb == 0 and lazy_b != lazy_a:
                    continue
                xa = _lazy_ref(a_raw.double(), na) if lazy_a else a_raw.double()
                x = xa
                if Cb:
                    x = torch.cat([xa, _lazy_ref(b_raw.double(), nb) if lazy_b else b_raw.double()], 1)
                ref = Fn.conv2d(x, w.double(), padding=1)
                y, norm = ops.unet_conv3x3((a_raw, na) if lazy_a else a_raw, None if not Cb else ((b_raw, nb) if lazy_b else b_raw), w)
                assert rel_l2(y, ref) <= 2e-6, (shape, f16, lazy_a, lazy_b, rel_l2(y, ref))
                mean, var = ref.mean((2, 3)), ref.var((2, 3), unbiased=False)
                assert (norm[..., 0].double() - mean).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max()))
                assert rel_l2(norm[..., 1], 1.0 / torch.sqrt(var + 1e-5)) <= 1e-5
                assert rel_l2(ops.unet_apply((y, norm)), _norm_act(ref)) <= 5e-6
    finally:
        ops.UNET_F16 = keep


def test_unet_conv3x3_two_term_fp16_scales(dev):
    """mrx_unet_conv3x3_h: fp32-level results for any magnitude of the inputs and weights (the block scales are powers of two taken from the
    bound of max |x| and from max |w|), with a measured bound (no attribute) and with an attached one that is 1000 x too large; a hot pixel
    1e4 x the rest costs the OTHER pixels nothing beyond 2^-39 of the bound (absolute)."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    B, C, Cout, H, W = 2, 14, 28, 37, 75
    for xs, ws in ((1.0, 1.0), (1e-4, 30.0), (3e4, 1e-3)):
        x, w = r(B, C, H, W) * xs, r(Cout, C, 3, 3) / (9 * C) ** 0.5 * ws
        ref = Fn.conv2d(x.double(), w.double(), padding=1)
        y, _ = ops.unet_conv3x3(x, None, w)                                   # bound measured (mrx_max_abs)
        assert rel_l2(y, ref) <= 1e-6, (xs, ws, rel_l2(y, ref))
        xb = x.clone()
        ops._attach_bound(xb, (x.abs().max() * 1000.0).reshape(1))            # a bound 1000 x too large: 10 of the 17 spare bits used up
        y2, _ = ops.unet_conv3x3(xb, None, w)
        assert rel_l2(y2, ref) <= 1e-6, (xs, ws, rel_l2(y2, ref))
        exact = ops.UNET_F16
        try:
            ops.UNET_F16 = False
            y32, _ = ops.unet_conv3x3(x, None, w)
        finally:
            ops.UNET_F16 = exact
        assert rel_l2(y, ref) <= 2.5 * rel_l2(y32, ref) + 1e-7                 # the fp32-input MFMA kernel's own distance from float64
    x, w = r(1, C, H, W), r(Cout, C, 3, 3) / (9 * C) ** 0.5
    x[:, :, 20, 40] *= 1e4
    ref = Fn.conv2d(x.double(), w.double(), padding=1)
    y, _ = ops.unet_conv3x3(x, None, w)
    far = torch.ones(1, Cout, H, W, dtype=torch.bool)
    far[:, :, 19:22, 39:42] = False
    d, rr = (y.double().cpu() - ref.cpu())[far], ref.cpu()[far]
    assert float((d.abs() / (rr.abs() + float(rr.pow(2).mean().sqrt()))).max()) <= 1e-5


@pytest.mark.parametrize("shape", [(1, 56, 28, 160, 95), (1, 28, 14, 320, 190), (2, 36, 18, 20, 12), (1, 6, 4, 5, 3), (1, 144, 72, 80, 48)])
def test_unet_transposed_conv_pool_and_1x1_read_lazy_tensors(shape, dev):
    """mrx_unet_conv_transpose2x2 (every channel-group width), mrx_unet_avgpool, mrx_unet_conv1x1 on plain and lazy inputs against float64."""
    from mridc_amd import ops
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    raw = r(B, Cin, H, W) * 1.5 + 0.2
    nrm = torch.stack([r(B, Cin) * 0.5, r(B, Cin).abs() + 0.5], -1)
    wt, w1, b1 = r(Cin, Cout, 2, 2) / (4 * Cin) ** 0.5, r(2, Cin, 1, 1) / Cin ** 0.5, r(2) * 0.1
    for lazy in (False, True):
        src = (raw, nrm) if lazy else raw
        x = _lazy_ref(raw.double(), nrm) if lazy else raw.double()
        ref = Fn.conv_transpose2d(x, wt.double(), stride=2)
        y, norm = ops.unet_conv_transpose2x2(src, wt)
        assert rel_l2(y, ref) <= 2e-6
        assert rel_l2(ops.unet_apply((y, norm)), _norm_act(ref)) <= 5e-6
        if H >= 2 and W >= 2:
            assert rel_l2(ops.unet_avg_pool2x2(src), Fn.avg_pool2d(x, 2)) <= 1e-6
        assert rel_l2(ops.unet_conv1x1(src, w1, b1), Fn.conv2d(x, w1.double(), b1.double())) <= 2e-6
        assert rel_l2(ops.unet_conv1x1(src, w1, None), Fn.conv2d(x, w1.double())) <= 2e-6


@pytest.mark.parametrize("cfg", [(14, 2, 11, 640, 372), (18, 4, 15, 640, 372), (8, 3, 7, 45, 37), (6, 2, 3, 33, 26)])
def test_norm_unet_fused_matches_conv_plus_apply(cfg, dev):
    """The whole NormUnet on the fused kernels against the conv + apply formulation (itself pinned by the reference goldens G7 and the
    full-size E2EVN test): the E2EVN shapes and odd sizes that take the reflect-pad fallback (unet_block.py:215-222)."""
    from mridc_amd.collections.reconstruction.models.unet_base.unet_block import NormUnet, Unet
    chans, pools, pad, H, W = cfg
    torch.manual_seed(chans + pools)
    net = NormUnet(chans, pools, padding_size=pad).eval().to(dev)
    x = torch.randn(1, 1, H, W, 2, generator=torch.Generator().manual_seed(H)).to(dev)
    keep = Unet.fused
    try:
        with torch.no_grad():
            Unet.fused = True
            got = net(x)
            Unet.fused = False
            want = net(x)
    finally:
        Unet.fused = keep
    assert rel_l2(got, want) <= 2e-5, rel_l2(got, want)


@pytest.mark.parametrize("shape", [(1, 1, 640, 372, 11), (2, 2, 45, 37, 7), (1, 1, 33, 26, 15), (3, 1, 16, 16, 3)])
def test_norm_unet_head_and_tail_in_one_pass(shape, dev):
    """mrx_unet_cnorm_pad against pad(norm(complex_to_chan_dim(x))) (unet_block.py:46-112: unbiased std per (batch, component)) and
    mrx_unet_conv1x1_cunnorm against chan_complex_to_last_dim(unnorm(unpad(conv1x1(.)))), both in float64."""
    import math
    from mridc_amd import ops
    B, c, H, W, ps = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(B, c, H, W, 2, generator=g) * 3 + torch.tensor([0.7, -1.1])).to(dev)
    w_mult, h_mult = ((W - 1) | ps) + 1, ((H - 1) | ps) + 1
    w_pad = [math.floor((w_mult - W) / 2), math.ceil((w_mult - W) / 2)]
    h_pad = [math.floor((h_mult - H) / 2), math.ceil((h_mult - H) / 2)]
    out, mean, std = ops.unet_cnorm_pad(x, h_pad, w_pad)
    xc = x.double().permute(0, 4, 1, 2, 3).reshape(B, 2 * c, H, W)
    v = xc.reshape(B, 2, c * H * W)
    m, sd = v.mean(2).view(B, 2, 1, 1), v.std(2).view(B, 2, 1, 1)
    ref = ((xc.view(B, 2, c, H * W) - m) / sd).view(B, 2 * c, H, W)
    ref = Fn.pad(ref, w_pad + h_pad)
    assert rel_l2(out, ref) <= 2e-6 and rel_l2(mean.view(B, 2), m.view(B, 2)) <= 2e-6 and rel_l2(std.view(B, 2), sd.view(B, 2)) <= 2e-6
    Cin = 5
    raw = torch.randn(B, Cin, h_mult, w_mult, generator=g).to(dev)
    nrm = torch.stack([torch.randn(B, Cin, generator=g) * 0.3, torch.rand(B, Cin, generator=g) + 0.5], -1).to(dev)
    w1, b1 = (torch.randn(2 * c, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev), (torch.randn(2 * c, generator=g) * 0.1).to(dev)
    for lazy in (True, False):
        xin = _lazy_ref(raw.double(), nrm) if lazy else raw.double()
        y = Fn.conv2d(xin, w1.double(), b1.double())[:, :, h_pad[0]:h_pad[0] + H, w_pad[0]:w_pad[0] + W]
        y = (y.reshape(B, 2, c * H * W) * sd.view(B, 2, 1) + m.view(B, 2, 1)).view(B, 2, c, H, W).permute(0, 2, 3, 4, 1)
        got = ops.unet_conv1x1_cunnorm((raw, nrm) if lazy else raw, w1, b1, mean, std, h_pad[0], w_pad[0], H, W)
        assert rel_l2(got, y) <= 2e-6


@pytest.mark.parametrize("cfg", [(14, 2, 11, 640, 372), (18, 4, 15, 320, 186), (8, 3, 7, 45, 37)])
def test_norm_unet_two_term_fp16_against_the_fp32_input_kernels(cfg, dev):
    """The whole NormUnet with its 3x3 convolutions on two-term fp16 operands (default) against the fp32-input MFMA kernels (ops.UNET_F16 = False)
    -- and, through MRIDC_AMD_ARITH=bf16x3, the same switch by the library's one arithmetic variable: ten normalised layers deep, 2e-5."""
    import os
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.unet_base.unet_block import NormUnet
    chans, pools, pad, H, W = cfg
    torch.manual_seed(chans + pools + 1)
    net = NormUnet(chans, pools, padding_size=pad).eval().to(dev)
    x = torch.randn(2, 1, H, W, 2, generator=torch.Generator().manual_seed(W)).to(dev)
    keep, env = ops.UNET_F16, os.environ.get("MRIDC_AMD_ARITH")
    try:
        with torch.no_grad():
            ops.UNET_F16 = True
            got = net(x)
            ops.UNET_F16 = False
            want = net(x)
            ops.UNET_F16 = True
            os.environ["MRIDC_AMD_ARITH"] = "bf16x3"
            want2 = net(x)
    finally:
        ops.UNET_F16 = keep
        if env is None:
            os.environ.pop("MRIDC_AMD_ARITH", None)
        else:
            os.environ["MRIDC_AMD_ARITH"] = env
    assert torch.equal(want, want2)                      # both switches select the same kernels
    assert rel_l2(got, want) <= 2e-5, rel_l2(got, want)


@pytest.mark.parametrize("shape", [(8, 14, 0, 14, 640, 380), (2, 14, 14, 28, 320, 190), (3, 5, 3, 36, 9, 7), (1, 1, 0, 1, 1, 2), (2, 28, 28, 70, 160, 95)])
def test_unet_conv3x3_statistics_merged_inside_the_launch(shape, dev):
    """mrx_unet_conv3x3_hc (round 5): the last tile of every plane -- found by a ticket per plane -- merges the plane's tile statistics itself, no
    k_unorm_finalize launch behind the convolution.  Same raw output bit for bit, (mean, 1/std) equal to the separate launch's up to the order of three
    double-precision sums; the ticket buffer is left zeroed (the next call on the stream reuses it), also when two streams run at once on their own buffers."""
    from mridc_amd import ops
    B, Ca, Cb, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    xa, xb = r(B, Ca, H, W) * 2 + 0.5, (r(B, Cb, H, W) - 0.3 if Cb else None)
    w = r(Cout, Ca + Cb, 3, 3) / (9 * (Ca + Cb)) ** 0.5
    keep = ops.UNET_FOLD_FINALIZE
    try:
        ops.UNET_FOLD_FINALIZE = False
        y0, n0 = ops.unet_conv3x3(xa, xb, w)
        ops.UNET_FOLD_FINALIZE = True
        for rep in range(3):                                   # the same ticket buffer three times
            y1, n1 = ops.unet_conv3x3(xa, xb, w)
            assert torch.equal(y1, y0)
            assert float((n1.double() - n0.double()).abs().max()) <= 1e-6 * max(1.0, float(n0.abs().max())), rep
        tk = ops._unet_tickets(B * Cout * 32, dev)
        torch.cuda.synchronize()
        assert int(tk.abs().sum()) == 0
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for st in (s1, s2, s1, s2):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(ops.unet_conv3x3(xa, xb, w))
        torch.cuda.synchronize()
        for y2, n2 in outs:
            assert torch.equal(y2, y0) and torch.equal(n2, n1)          # (the merge itself is deterministic: fixed order, whoever runs it)
    finally:
        ops.UNET_FOLD_FINALIZE = keep


def test_statistics_merged_inside_the_launch_under_stress(dev):
    """The ticket that announces a tile's statistics must not overtake them (round-5 advisor finding: the write-through stores are now acknowledged --
    s_waitcnt vmcnt(0) in the storing thread -- before the ticket is taken).  GPU sanitizers are not available on this pool: many small planes (tiles of one
    plane spread over all XCDs, merges by whichever workgroup comes last), 200 launches back to back, every (mean, 1/std) pair compared with the separate
    finalize launch's -- a stale statistic is an error of the size of the statistic, not of a rounding."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(99)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    keep = ops.UNET_FOLD_FINALIZE
    try:
        for B, C, H, W in ((24, 14, 40, 72), (6, 28, 96, 96), (64, 14, 8, 33)):
            x, w = r(B, C, H, W) * 3 + 1.0, r(C, C, 3, 3) / (9 * C) ** 0.5
            ops.UNET_FOLD_FINALIZE = False
            y0, n0 = ops.unet_conv3x3(x, None, w)
            ops.UNET_FOLD_FINALIZE = True
            tol = 1e-6 * max(1.0, float(n0.abs().max()))
            bad = 0
            for rep in range(200 // (1 if B < 64 else 2)):
                y1, n1 = ops.unet_conv3x3(x, None, w)
                bad += int(((n1.double() - n0.double()).abs() > tol).sum())
            assert bad == 0, (B, C, H, W, bad)
            assert torch.equal(y1, y0)
    finally:
        ops.UNET_FOLD_FINALIZE = keep

