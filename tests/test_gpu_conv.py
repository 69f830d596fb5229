"""GPU parity: the MFMA convolution / recurrent-cell / NormUnet kernels against plain torch fp32 CPU references
(the oracle's building blocks).  Tolerance: rel-L2 <= 1e-5 (fp32 fma chains in a different summation order)."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests._util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


CASES = [  # B, Cin, Cout, H, W, k, dil
    (1, 4, 64, 16, 12, 5, 1), (2, 64, 64, 13, 18, 3, 2), (1, 64, 64, 40, 70, 3, 2), (1, 3, 5, 9, 33, 3, 1),
    (2, 7, 33, 17, 19, 3, 1), (1, 2, 14, 32, 16, 3, 1), (1, 28, 14, 15, 12, 3, 1), (1, 16, 48, 16, 12, 3, 1),
    (1, 64, 64, 8, 32, 1, 1), (1, 5, 96, 10, 37, 5, 2), (1, 1, 1, 1, 1, 3, 1), (1, 18, 130, 9, 9, 3, 1),
    # NormUnet shapes: the tuned 3x3 kernel (16-wide MFMA blocks, LDS-DMA tile), 1..4 cout blocks, ragged tiles, batch
    (1, 14, 14, 24, 100, 3, 1), (1, 56, 56, 20, 24, 3, 1), (1, 56, 28, 17, 21, 3, 1), (2, 28, 28, 9, 95, 3, 1),
    (1, 14, 28, 33, 47, 3, 1), (1, 2, 14, 64, 380, 3, 1), (1, 64, 64, 8, 32, 3, 1), (3, 6, 40, 5, 7, 3, 1),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("pad_mode", ["zero", "replicate"])
def test_conv2d_vs_torch(case, pad_mode, dev):
    from mridc_amd import ops
    B, Cin, Cout, H, W, k, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    p = dil * (k - 1) // 2
    if pad_mode == "replicate":
        ref = F.conv2d(F.pad(x, (p, p, p, p), mode="replicate") if p else x, w, b, dilation=dil)
        got = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), dil, ops.PAD_REPLICATE, ops.ACT_RELU)
        ref = F.relu(ref)
    else:
        ref = F.leaky_relu(F.conv2d(x, w, None, padding=p, dilation=dil), 0.2)
        got = ops.conv2d(x.to(dev), w.to(dev), None, dil, ops.PAD_ZERO, ops.ACT_LEAKY, 0.2)
    assert_close(got, ref, 1e-5, f"conv2d {case} {pad_mode}")


@pytest.mark.parametrize("shape", [(1, 14, 14, 24, 100), (2, 2, 14, 64, 380), (1, 56, 56, 20, 24), (1, 28, 56, 17, 21), (2, 28, 28, 9, 95),
                                   (1, 14, 28, 33, 47), (1, 6, 40, 5, 7), (1, 8, 100, 12, 12), (1, 3, 9, 1, 2)])
def test_conv_instance_norm_fused_statistics(shape, dev):
    """Conv3x3 -> InstanceNorm2d -> LeakyReLU(0.2) with the statistics taken from the conv accumulators (per-tile mean / M2 merged
    with the parallel-variance formula) against torch's conv + instance_norm (which accumulates its statistics in double)."""
    from mridc_amd import ops
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g) + 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5 + 0.05       # non-zero output mean: M2 is not just E[x^2]
    ref = F.leaky_relu(F.instance_norm(F.conv2d(x, w, None, padding=1), eps=1e-5), 0.2)
    got = ops.conv_instance_norm_act(x.to(dev), w.to(dev), 1e-5, ops.ACT_LEAKY, 0.2)
    assert_close(got, ref, 1e-5, f"conv + instance norm {shape}")
    sep = ops.instance_norm_act(ops.conv2d(x.to(dev), w.to(dev), None, 1, ops.PAD_ZERO), 1e-5, ops.ACT_LEAKY, 0.2)
    assert_close(got, sep, 5e-6, "fused statistics vs the three-pass instance norm")
    # (Cout > 64 -- the (18, 4) U-Net reaches 288 channels -- runs the same tuned kernel with its cout blocks spread over grid.y)
    # the two-launch form of the same thing: merged statistics as a tensor, then the apply pass
    y, stats = ops.conv2d_stats(x.to(dev), w.to(dev))
    conv = F.conv2d(x, w, None, padding=1).double()
    assert_close(stats[..., 0].cpu(), conv.mean((2, 3)).float(), 1e-5, "plane means")
    assert_close(stats[..., 1].cpu(), ((conv - conv.mean((2, 3), keepdim=True)) ** 2).sum((2, 3)).float(), 1e-5, "plane M2")
    assert_close(ops.instance_norm_apply(y, stats, 1e-5, ops.ACT_LEAKY, 0.2), got, 1e-6, "apply from merged statistics")


def test_conv2d_small_cout_and_identity(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(1)
    for (Cin, Cout, k) in ((14, 2, 1), (64, 2, 3), (8, 4, 3), (6, 1, 5), (5, 3, 3)):
        x = torch.randn(2, Cin, 11, 35, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g)
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, b, padding=(k - 1) // 2)
        assert_close(ops.conv2d(x.to(dev), w.to(dev), b.to(dev), 1, ops.PAD_ZERO), ref, 1e-5, f"small conv {Cin}->{Cout}")
    # transpose-detecting check: identity kernel with an asymmetric input must return the input
    x = torch.arange(2 * 33 * 9 * 40, dtype=torch.float32).reshape(2, 33, 9, 40) / 1000
    w = torch.zeros(33, 33, 3, 3)
    for c in range(33):
        w[c, c, 1, 1] = 1.0
    assert_close(ops.conv2d(x.to(dev), w.to(dev), None, 1, ops.PAD_ZERO), x, 1e-7, "identity conv")


@pytest.mark.parametrize("F_", [32, 64])
@pytest.mark.parametrize("shape", [(1, 4, 16, 12, 5, 1), (2, 64, 13, 18, 3, 2), (1, 64, 24, 70, 3, 2)])
def test_fused_rim_layer_vs_unfused_reference(F_, shape, dev):
    from mridc_amd import ops
    B, Cin, H, W, k, dil = shape
    g = torch.Generator().manual_seed(F_ + sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    wc = torch.randn(F_, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bc = torch.randn(F_, generator=g) * 0.1
    wi = torch.randn(F_, F_, 1, 1, generator=g) / F_ ** 0.5
    bi = torch.randn(F_, generator=g) * 0.1
    hh = torch.randn(1, F_, 1, 1, generator=g) * 0.5
    hp = torch.randn(B, F_, H, W, generator=g)
    ref = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, bc, k, dil, "relu"), hp, wi, bi, hh, 1, 1)
    got = ops.rim_layer_indrnn(x.to(dev), wc.to(dev), bc.to(dev), k, dil, wi.to(dev), bi.to(dev), hh.to(dev), hp.to(dev))
    assert_close(got, ref, 1e-5, f"fused layer F={F_} {shape}")
    ref0 = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, None, k, dil, "relu"), torch.zeros_like(hp), wi, None, hh, 1, 1)
    got0 = ops.rim_layer_indrnn(x.to(dev), wc.to(dev), None, k, dil, wi.to(dev), None, hh.to(dev), None)
    assert_close(got0, ref0, 1e-5, "fused layer, no biases, h_prev = None")


@pytest.mark.parametrize("shape", [(1, 4, 16, 12, 5, 1), (2, 64, 13, 18, 3, 2), (1, 64, 24, 70, 3, 2), (1, 16, 9, 33, 3, 1),
                                   (1, 64, 8, 32, 1, 1), (1, 7, 17, 19, 3, 2), (2, 3, 1, 1, 5, 1), (1, 12, 40, 100, 5, 1)])
def test_tuned_fused_rim_layer_packed(shape, dev):
    """The tuned kernel (pre-packed weights, 8 waves, register-chained 1x1) against the oracle and the generic kernel."""
    from mridc_amd import ops
    B, Cin, H, W, k, dil = shape
    F_ = 64
    assert ops.rim_layer_supported(Cin, F_, k, dil)
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    wc = torch.randn(F_, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bc = torch.randn(F_, generator=g) * 0.1
    wi = torch.randn(F_, F_, 1, 1, generator=g) / F_ ** 0.5
    bi = torch.randn(F_, generator=g) * 0.1
    hh = torch.randn(1, F_, 1, 1, generator=g) * 0.5
    hp = torch.randn(B, F_, H, W, generator=g)
    ref = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, bc, k, dil, "relu"), hp, wi, bi, hh, 1, 1)
    packed = ops.rim_layer_pack(wc.to(dev), wi.to(dev))
    got = ops.rim_layer_indrnn_packed(x.to(dev), packed, F_, k, dil, bc.to(dev), bi.to(dev), hh.to(dev), hp.to(dev))
    assert_close(got, ref, 1e-5, f"tuned fused layer {shape}")
    gen = ops.rim_layer_indrnn(x.to(dev), wc.to(dev), bc.to(dev), k, dil, wi.to(dev), bi.to(dev), hh.to(dev), hp.to(dev))
    assert_close(got, gen, 2e-6, "tuned vs generic kernel")
    ref0 = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, None, k, dil, "relu"), torch.zeros_like(hp), wi, None, hh, 1, 1)
    got0 = ops.rim_layer_indrnn_packed(x.to(dev), packed, F_, k, dil, None, None, hh.to(dev), None)
    assert_close(got0, ref0, 1e-5, "tuned fused layer, no biases, h_prev = None")


@pytest.mark.parametrize("shape", [(1, 640, 372), (2, 37, 75), (1, 19, 33), (3, 16, 32), (1, 5, 3), (1, 130, 320), (1, 33, 64)])
def test_second_rim_layer_split_bf16_has_fp32_accuracy(shape, dev):
    """The dominant layer (3x3 dilation 2, 64 -> 64, + IndRNN 1x1) as a direct convolution on the bf16 matrix pipe (k_rim_layer2_sb: three-term
    bf16 operand split, six term products per multiply) against a float64 reference: fp32-level error, and agreement with the fp32 Winograd
    kernel to fp32 round-off.  Ragged tiles (H % 16, W % 32), images smaller than a tile, batches, no h_prev, no biases."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, H, W = shape
    F_ = 64
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, F_, H, W, generator=g).relu()
    wc = torch.randn(F_, F_, 3, 3, generator=g) / 24
    bc = torch.randn(F_, generator=g) * 0.1
    wi = torch.randn(F_, F_, 1, 1, generator=g) / 8
    bi = torch.randn(F_, generator=g) * 0.1
    hh = torch.randn(1, F_, 1, 1, generator=g) * 0.5
    hp = torch.randn(B, F_, H, W, generator=g).relu()
    pk_s, pk_w = ops.rim_layer2_sb_pack(wc.to(dev), wi.to(dev)), ops.rim_layer_wino_pack(wc.to(dev), wi.to(dev))
    for with_state, with_bias in ((True, True), (False, True), (True, False)):
        b1, b2 = (bc, bi) if with_bias else (None, None)
        ref = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), None if b1 is None else b1.double(), dilation=2).relu()
        ref = Fn.conv2d(ref, wi.double(), None if b2 is None else b2.double())
        ref = Fn.relu(ref + hh.double() * hp.double() if with_state else ref)
        d = lambda t: None if t is None else t.to(dev)  # noqa: E731
        got = ops.rim_layer2_sb(x.to(dev), pk_s, d(b1), d(b2), hh.to(dev), hp.to(dev) if with_state else None)
        win = ops.rim_layer_indrnn_wino(x.to(dev), pk_w, F_, d(b1), d(b2), hh.to(dev), hp.to(dev) if with_state else None)
        e_sb, e_w = rel_l2(got, ref), rel_l2(win, ref)
        assert e_sb <= 6e-7 and e_sb <= 2.5 * e_w + 5e-8, (e_sb, e_w)
        assert rel_l2(got, win) <= 8e-7


@pytest.mark.parametrize("shape", [(1, 640, 372), (2, 37, 75), (1, 19, 33), (3, 16, 32), (1, 5, 3), (1, 1, 1), (1, 130, 320)])
def test_second_rim_layer_with_final_conv_in_its_tail(shape, dev):
    """mrx_rim_layer2_sb_final: the layer's h_new must be the plain kernel's bit for bit, and eta + permute(conv3x3_reppad(h_new) + b) must
    match a float64 convolution of that h_new at fp32 round-off and the stand-alone final kernel (mrx_rim_final) to round-off (rim_block.py:233-246).
    Ragged tiles, images smaller than a tile (every tap clamped), batches, no h_prev, no biases."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, H, W = shape
    F_ = 64
    g = torch.Generator().manual_seed(7 + sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(B, F_, H, W).relu(), r(B, F_, H, W).relu()
    wc, wi, wf = r(F_, F_, 3, 3) / 24, r(F_, F_, 1, 1) / 8, r(2, F_, 3, 3) / 24
    bc, bi, bf, hh = r(F_) * 0.1, r(F_) * 0.1, r(2) * 0.1, r(1, F_, 1, 1) * 0.5
    eta = r(B, H, W, 2)
    pk = ops.rim_layer2_sb_pack(wc, wi, wf)
    for with_state, with_bias in ((True, True), (False, True), (True, False)):
        b1, b2, b3 = (bc, bi, bf) if with_bias else (None, None, None)
        h_plain = ops.rim_layer2_sb(x, pk, b1, b2, hh, hp if with_state else None)
        h_new, eta_new = ops.rim_layer2_sb_final(x, pk, b1, b2, hh, hp if with_state else None, b3, eta)
        assert torch.equal(h_new, h_plain)
        conv = Fn.conv2d(Fn.pad(h_new.double(), (1, 1, 1, 1), mode="replicate"), wf.double(), None if b3 is None else b3.double())
        ref = eta.double() + conv.permute(0, 2, 3, 1)
        sep = ops.rim_final(h_new, wf, b3, 3, 1, eta)
        e_f, e_s = rel_l2(eta_new, ref), rel_l2(sep, ref)
        assert e_f <= 4e-7 and e_f <= 2.5 * e_s + 5e-8, (e_f, e_s)
        # (the update is a difference of fp32 results: with a handful of values one ulp of eta against the size of the update shows)
        assert rel_l2(eta_new - eta, sep - eta) <= (1e-6 if H * W >= 16 else 3e-6)


@pytest.mark.parametrize("shape", [(1, 8, 128, 5, 256, 256), (1, 2, 64, 3, 640, 372), (2, 4, 64, 5, 37, 75), (1, 1, 32, 3, 19, 33), (3, 7, 100, 5, 8, 32),
                                   (1, 3, 33, 3, 5, 3), (1, 8, 128, 3, 1, 1)])
def test_few_channel_conv_split_bf16(shape, dev):
    """mrx_conv_sbs (3x3 / 5x5 convolutions of <= 8 channels into <= 128: the cascades' first layers, e.g. qRIM's 5x5 8 -> 128 + ReLU) through
    ops.conv2d against float64 and the generic fp32-MFMA kernel it replaces: fp32-level error; zero and replicate padding, every activation,
    ragged tiles, channel counts off the block sizes."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, Cin, Cout, k, H, W = shape
    g = torch.Generator().manual_seed(3 + sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, w, b = r(B, Cin, H, W), r(Cout, Cin, k, k) / (Cin * k * k) ** 0.5, r(Cout) * 0.1
    assert ops.conv_sbs_supported(Cin, Cout, k, 1)
    keep = ops.SBS_CONV
    try:
        for pm, mode in ((ops.PAD_ZERO, "constant"), (ops.PAD_REPLICATE, "replicate")):
            for act, bias in ((ops.ACT_NONE, b), (ops.ACT_RELU, None), (ops.ACT_LEAKY, b)):
                ref = Fn.conv2d(Fn.pad(x.double(), (k // 2,) * 4, mode=mode), w.double(), None if bias is None else bias.double())
                ref = ref.relu() if act == ops.ACT_RELU else (Fn.leaky_relu(ref, 0.1) if act == ops.ACT_LEAKY else ref)
                got = ops.conv_sbs(x, w, bias, pm, act, 0.1)          # (ops.conv2d routes the 5x5 shapes here)
                if k == 5 and Cout >= ops.SBS_MIN_COUT:
                    ops.SBS_CONV = True
                    assert torch.equal(ops.conv2d(x, w, bias, 1, pm, act, 0.1), got)
                ops.SBS_CONV = False
                old = ops.conv2d(x, w, bias, 1, pm, act, 0.1)
                e_sb, e_old = rel_l2(got, ref), rel_l2(old, ref)
                assert e_sb <= 5e-7 and e_sb <= 2.5 * e_old + 5e-8, (shape, mode, act, e_sb, e_old)
    finally:
        ops.SBS_CONV = keep


@pytest.mark.parametrize("shape", [(1, 640, 372), (2, 37, 75), (1, 19, 33), (3, 16, 32), (1, 5, 3), (1, 1, 1), (1, 130, 320)])
def test_conv3x3_64_to_64_split_bf16(shape, dev):
    """mrx_conv3x3_sb (the convolution stage of the dominant RIM layer as a general 64 -> 64 convolution: dilation 1 and 2, zero and replicate
    padding, bias, none / ReLU / LeakyReLU) against float64 and the fp32 Winograd kernel it replaces in ops.conv2d: fp32-level error."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, H, W = shape
    g = torch.Generator().manual_seed(11 + sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, w, b = r(B, 64, H, W), r(64, 64, 3, 3) / 24, r(64) * 0.1
    keep = ops.SB_CONV
    try:
        for dil in (1, 2):
            for pm, mode in ((ops.PAD_ZERO, "constant"), (ops.PAD_REPLICATE, "replicate")):
                for act, bias in ((ops.ACT_NONE, b), (ops.ACT_RELU, None), (ops.ACT_LEAKY, b)):
                    ref = Fn.conv2d(Fn.pad(x.double(), (dil,) * 4, mode=mode), w.double(), None if bias is None else bias.double(), dilation=dil)
                    ref = ref.relu() if act == ops.ACT_RELU else (Fn.leaky_relu(ref, 0.1) if act == ops.ACT_LEAKY else ref)
                    ops.SB_CONV = True
                    got = ops.conv2d(x, w, bias, dil, pm, act, 0.1)
                    ops.SB_CONV = False
                    old = ops.conv2d(x, w, bias, dil, pm, act, 0.1)
                    e_sb, e_old = rel_l2(got, ref), rel_l2(old, ref)
                    assert e_sb <= 6e-7 and e_sb <= 2.5 * e_old + 5e-8, (shape, dil, mode, act, e_sb, e_old)
    finally:
        ops.SB_CONV = keep


@pytest.mark.parametrize("shape", [(1, 128, 4, 256, 256), (1, 64, 2, 640, 372), (2, 64, 3, 37, 29), (1, 128, 1, 5, 3), (3, 64, 4, 1, 1)])
def test_thin_3x3_conv_as_contraction_plus_gather(shape, dev):
    """ops.conv2d for 3x3 convolutions of 64 / 128 channels into <= 4 (qRIM's final layer, qrim_block.py:226-236 via conv_layers.py:121-123):
    a 1x1 channel contraction into 9 Cout tap planes on the matrix cores + mrx_taps_gather, against float64 and the direct kernel, zero and
    replicate padding, images smaller than the 3x3 window."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, w, b = r(B, Cin, H, W), r(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5, r(Cout) * 0.1
    assert ops.conv3x3_taps_supported(Cin, Cout)
    keep = ops.TAPS_CONV
    try:
        for pm, mode in ((ops.PAD_ZERO, "constant"), (ops.PAD_REPLICATE, "replicate")):
            for bias in (b, None):
                ref = Fn.conv2d(Fn.pad(x.double(), (1, 1, 1, 1), mode=mode), w.double(), None if bias is None else bias.double())
                ops.TAPS_CONV = True
                got = ops.conv2d(x, w, bias, 1, pm)
                ops.TAPS_CONV = False
                direct = ops.conv2d(x, w, bias, 1, pm)
                assert rel_l2(got, ref) <= 1e-6 and rel_l2(got, direct) <= 2e-6, (shape, mode, rel_l2(got, ref), rel_l2(got, direct))
    finally:
        ops.TAPS_CONV = keep


@pytest.mark.parametrize("shape", [(1, 4, 640, 372), (2, 4, 37, 75), (1, 2, 19, 33), (3, 1, 16, 32), (1, 4, 5, 3), (1, 3, 130, 320)])
def test_first_rim_layer_split_bf16_has_fp32_accuracy(shape, dev, monkeypatch):
    """The first RIM layer on the bf16 matrix pipe (k_rim_layer1_sb: every fp32 operand as the exact sum of three bf16 terms, six term
    products per multiply) against a float64 reference: its error must be that of the fp32-MFMA kernel (MRIDC_AMD_ARITH=fp32), and the two
    kernels must agree to fp32 round-off.  Ragged tiles (W % 32, H % 16), fewer than four input channels, batches, no h_prev."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, Cin, H, W = shape
    F_ = 64
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    wc = torch.randn(F_, Cin, 5, 5, generator=g) / (Cin * 25) ** 0.5
    bc = torch.randn(F_, generator=g) * 0.1
    wi = torch.randn(F_, F_, 1, 1, generator=g) / F_ ** 0.5
    bi = torch.randn(F_, generator=g) * 0.1
    hh = torch.randn(1, F_, 1, 1, generator=g) * 0.5
    hp = torch.randn(B, F_, H, W, generator=g).relu()
    packed = ops.rim_layer_pack(wc.to(dev), wi.to(dev))
    for with_state in (True, False):
        ref = Fn.relu(Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double()))
        ref = Fn.conv2d(ref, wi.double(), bi.double())
        ref = Fn.relu(ref + hh.double() * hp.double() if with_state else ref)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("MRIDC_AMD_ARITH", "fp32" if mode == "1" else "bf16x3")
            out[mode] = ops.rim_layer_indrnn_packed(x.to(dev), packed, F_, 5, 1, bc.to(dev), bi.to(dev), hh.to(dev),
                                                    hp.to(dev) if with_state else None).cpu()
        e_sb, e_fp = rel_l2(out["0"], ref), rel_l2(out["1"], ref)
        assert e_fp <= 4e-7 and e_sb <= 4e-7 and e_sb <= 1.5 * e_fp + 2e-8, (e_sb, e_fp)
        assert rel_l2(out["0"], out["1"]) <= 5e-7


@pytest.mark.parametrize("shape", [(2, 64, 13, 18), (1, 64, 24, 70), (1, 7, 17, 19), (1, 64, 8, 32), (3, 20, 1, 1), (1, 64, 40, 128),
                                   (1, 9, 2, 3), (1, 64, 33, 61), (1, 64, 24, 72), (2, 16, 19, 36), (1, 8, 9, 4), (1, 12, 5, 100),
                                   (1, 64, 64, 372)])
def test_winograd_fused_rim_layer(shape, dev):
    """Winograd F(2x2,3x3) on the dilation-2 parity sub-lattices against the oracle and the direct tuned kernel.
    Tolerance: 1e-5 of the output norm against the oracle (same bar as the direct kernel), 5e-6 against the direct kernel."""
    from mridc_amd import ops
    B, Cin, H, W = shape
    F_, k, dil = 64, 3, 2
    assert ops.rim_layer_wino_supported(Cin, F_, k, dil)
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn(B, Cin, H, W, generator=g)
    wc = torch.randn(F_, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bc = torch.randn(F_, generator=g) * 0.1
    wi = torch.randn(F_, F_, 1, 1, generator=g) / F_ ** 0.5
    bi = torch.randn(F_, generator=g) * 0.1
    hh = torch.randn(1, F_, 1, 1, generator=g) * 0.5
    hp = torch.randn(B, F_, H, W, generator=g)
    ref = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, bc, k, dil, "relu"), hp, wi, bi, hh, 1, 1)
    packed = ops.rim_layer_wino_pack(wc.to(dev), wi.to(dev))
    got = ops.rim_layer_indrnn_wino(x.to(dev), packed, F_, bc.to(dev), bi.to(dev), hh.to(dev), hp.to(dev))
    assert_close(got, ref, 1e-5, f"winograd fused layer {shape}")
    direct = ops.rim_layer_indrnn_packed(x.to(dev), ops.rim_layer_pack(wc.to(dev), wi.to(dev)), F_, k, dil, bc.to(dev), bi.to(dev),
                                         hh.to(dev), hp.to(dev))
    assert_close(got, direct, 5e-6, "winograd vs direct kernel")
    ref0 = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(x, wc, None, k, dil, "relu"), torch.zeros_like(hp), wi, None, hh, 1, 1)
    got0 = ops.rim_layer_indrnn_wino(x.to(dev), packed, F_, None, None, hh.to(dev), None)
    assert_close(got0, ref0, 1e-5, "winograd fused layer, no biases, h_prev = None")


def test_cells_vs_oracle(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(2)
    B, Cin, Fh, H, W = 2, 16, 16, 13, 18
    x = torch.randn(B, Cin, H, W, generator=g)
    h = torch.randn(B, Fh, H, W, generator=g)
    for k, d in ((1, 1), (3, 1), (3, 2)):
        wi = torch.randn(Fh, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        bi = torch.randn(Fh, generator=g)
        hh = torch.randn(1, Fh, 1, 1, generator=g)
        assert_close(ops.indrnn_cell(x.to(dev), wi.to(dev), bi.to(dev), hh.to(dev), h.to(dev), d),
                     oracle.rim.indrnn_cell(x, h, wi, bi, hh, k, d), 1e-5, f"indrnn k={k} d={d}")
        w3 = torch.randn(3 * Fh, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b3 = torch.randn(3 * Fh, generator=g)
        u3 = torch.randn(3 * Fh, Fh, k, k, generator=g) / (Fh * k * k) ** 0.5
        ih = ops.conv2d(x.to(dev), w3.to(dev), b3.to(dev), d, ops.PAD_ZERO)
        hhc = ops.conv2d(h.to(dev), u3.to(dev), None, d, ops.PAD_ZERO)
        assert_close(ops.gru_gates(ih, hhc, h.to(dev)), oracle.rim.convgru_cell(x, h, w3, b3, u3, k, d), 1e-5, "gru")
        ih2 = ops.conv2d(x.to(dev), w3[:2 * Fh].to(dev), b3[:2 * Fh].to(dev), d, ops.PAD_ZERO)
        hh2 = ops.conv2d(h.to(dev), u3[:2 * Fh].to(dev), None, d, ops.PAD_ZERO)
        assert_close(ops.mgu_gates(ih2, hh2, h.to(dev)), oracle.rim.convmgu_cell(x, h, w3[:2 * Fh], b3[:2 * Fh], u3[:2 * Fh], k, d),
                     1e-5, "mgu")


@pytest.mark.parametrize("shape", [(2, 13, 18), (1, 33, 70), (1, 64, 32), (3, 1, 5)])
def test_gated_cell_1x1_vs_oracle(dev, shape):
    """One-launch ConvGRUCell / ConvMGUCell (1x1 gate kernels, 64 -> 64) against the oracle cells; pixel counts that are not
    multiples of the 32-pixel wave segment, with and without bias, zero initial state as h = None."""
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(11 + H)
    F_ = 64
    x, h = torch.randn(B, F_, H, W, generator=g), torch.randn(B, F_, H, W, generator=g)
    for gates, cell in ((3, oracle.rim.convgru_cell), (2, oracle.rim.convmgu_cell)):
        wi = torch.randn(gates * F_, F_, 1, 1, generator=g) / 8
        wh = torch.randn(gates * F_, F_, 1, 1, generator=g) / 8
        bi = torch.randn(gates * F_, generator=g)
        assert ops.gated_cell_supported(F_, F_, 1, gates) and not ops.gated_cell_supported(F_, F_, 3, gates)
        packed = ops.gated_cell_pack(wi.to(dev), wh.to(dev), gates)
        assert_close(ops.gated_cell_1x1(x.to(dev), h.to(dev), packed, bi.to(dev), gates), cell(x, h, wi, bi, wh, 1, 1), 1e-5,
                     f"gated cell gates={gates} {shape}")
        assert_close(ops.gated_cell_1x1(x.to(dev), None, packed, None, gates), cell(x, torch.zeros_like(h), wi, None, wh, 1, 1), 1e-5,
                     f"gated cell gates={gates} {shape}, no bias, h = None")
        # the unfused route (two convs + gate kernel) is the same function
        ih = ops.conv2d(x.to(dev), wi.to(dev), bi.to(dev), 1, ops.PAD_ZERO)
        hh = ops.conv2d(h.to(dev), wh.to(dev), None, 1, ops.PAD_ZERO)
        unfused = (ops.gru_gates if gates == 3 else ops.mgu_gates)(ih, hh, h.to(dev))
        assert_close(ops.gated_cell_1x1(x.to(dev), h.to(dev), packed, bi.to(dev), gates), unfused, 1e-5, "one launch vs unfused")
    with pytest.raises(ValueError):
        ops.gated_cell_pack(torch.randn(48, 16, 1, 1).to(dev), torch.randn(48, 16, 1, 1).to(dev), 3)


@pytest.mark.parametrize("layer", ["GRU", "MGU"])
def test_rimblock_gated64_vs_oracle(dev, layer):
    """RIMBlock with the model-zoo RIM layout (5x5 -> GRU, 3x3 d2 -> GRU, 64 features, 1x1 gates): conv as the fused layer kernel
    with an identity ih + the one-launch cell, against the oracle block; second call continues from the returned hx."""
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    cfg = dict(recurrent_layer=layer, conv_filters=[64, 64, 2], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1],
               conv_bias=[True, True, False], recurrent_filters=[64, 64, 0], recurrent_kernels=[1, 1, 0],
               recurrent_dilations=[1, 1, 0], recurrent_bias=[True, True, False], depth=2, time_steps=3, conv_dim=2, no_dc=True,
               fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1)
    torch.manual_seed(21)
    blk = RIMBlock(**cfg).eval()
    with torch.no_grad():
        for n_, p_ in blk.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0, 0.1)
    p = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 1, 3, 24, 36
    S = torch.randn(B, C, H, W, 2, generator=g) / C ** 0.5
    mask = (torch.rand(1, 1, 1, W, 1, generator=g) < 0.4)
    y = torch.randn(B, C, H, W, 2, generator=g) * mask
    ocfg = oracle.rim.RIMConfig(**{k: v for k, v in cfg.items()})
    ref, ref_hx = oracle.rim.rim_block_forward(p, ocfg, y, y, S, mask, None, None, 1.0, False)
    blk = blk.to(dev)
    assert all(RIMBlock._gated(st) for st in blk.layers), "the test must exercise the one-launch cell"
    with torch.no_grad():
        outs, hx = blk(y.to(dev), y.to(dev), S.to(dev), mask.to(dev), None, None, 1.0, False)
    assert_close(torch.stack(outs), torch.stack(ref), 2e-5, f"{layer}-64 RIMBlock outs")
    for j in range(2):
        assert_close(hx[j], ref_hx[j], 2e-5, f"{layer}-64 hx{j}")
    ref2, _ = oracle.rim.rim_block_forward(p, ocfg, ref, y, S, mask, ref[-1], ref_hx, 1.0, False)
    with torch.no_grad():
        outs2, _ = blk(outs, y.to(dev), S.to(dev), mask.to(dev), outs[-1], hx, 1.0, False)
    assert_close(torch.stack(outs2), torch.stack(ref2), 5e-5, f"{layer}-64 RIMBlock, second call with hx")


def test_rim_final_vs_oracle(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(4)
    for (B, Fh, H, W, k, d) in ((1, 64, 16, 12, 3, 1), (2, 16, 13, 37, 3, 1), (1, 8, 9, 9, 5, 2), (1, 20, 33, 70, 3, 1),
                                (1, 128, 17, 40, 3, 1), (1, 64, 8, 8, 1, 1), (1, 24, 12, 12, 5, 1),
                                # W % 4 == 0 and F % 8 == 0: the 4-pixels-per-thread kernel (k = 3), all border cases
                                (1, 64, 24, 72, 3, 1), (2, 8, 9, 8, 3, 1), (1, 64, 40, 372, 3, 1), (1, 16, 1, 32, 3, 1), (1, 32, 19, 100, 3, 1)):
        h = torch.randn(B, Fh, H, W, generator=g)
        w = torch.randn(2, Fh, k, k, generator=g) / (Fh * k * k) ** 0.5
        eta = torch.randn(B, H, W, 2, generator=g)
        bias = torch.randn(2, generator=g)
        ref = eta + oracle.rim.conv_nonlinear(h, w, None, k, d, None).permute(0, 2, 3, 1)
        assert_close(ops.rim_final(h.to(dev), w.to(dev), None, k, d, eta.to(dev)), ref, 1e-5, f"rim_final {(B, Fh, H, W, k, d)}")
        refb = eta + oracle.rim.conv_nonlinear(h, w, bias, k, d, None).permute(0, 2, 3, 1)
        assert_close(ops.rim_final(h.to(dev), w.to(dev), bias.to(dev), k, d, eta.to(dev)), refb, 1e-5, "rim_final with bias")


def test_unet_pieces_vs_torch(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 6, 17, 22, generator=g) * 3 + 1
    assert_close(ops.instance_norm_act(x.to(dev), 1e-5, ops.ACT_LEAKY, 0.2, inplace=False),
                 F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2), 1e-5, "instance norm + leaky")
    xn, mean, std = ops.group_norm(x.to(dev), 2)
    gx = x.reshape(2, 2, -1)
    assert_close(mean, gx.mean(-1, keepdim=True), 1e-6, "group mean")
    assert_close(std, gx.std(-1, keepdim=True), 1e-6, "group std (unbiased)")
    assert_close(xn, ((gx - gx.mean(-1, keepdim=True)) / gx.std(-1, keepdim=True)).reshape(x.shape), 1e-5, "group norm")
    assert_close(ops.group_unnorm(xn, mean, std, 2), x, 1e-5, "group unnorm")
    assert_close(ops.pad2d(x.to(dev), 2, 3, 1, 4, 0), F.pad(x, (1, 4, 2, 3)), 0.0 + 1e-12, "zero pad")
    assert_close(ops.pad2d(x.to(dev), 0, 1, 0, 1, 1), F.pad(x, (0, 1, 0, 1), "reflect"), 1e-12, "reflect pad")
    assert_close(ops.pad2d(ops.pad2d(x.to(dev), 2, 3, 1, 4, 0), -2, -3, -1, -4, 0), x, 1e-12, "unpad")
    assert_close(ops.avg_pool2x2(x.to(dev)), F.avg_pool2d(x, 2, 2), 1e-6, "avg pool")
    w = torch.randn(6, 4, 2, 2, generator=g)
    assert_close(ops.conv_transpose2x2(x.to(dev), w.to(dev)), F.conv_transpose2d(x, w, stride=2), 1e-5, "conv transpose")
    for (B_, Ci, Co, H_, W_) in ((1, 28, 14, 33, 47), (2, 56, 28, 9, 23), (1, 16, 8, 5, 70), (1, 7, 3, 4, 4), (1, 28, 14, 320, 190)):
        xt = torch.randn(B_, Ci, H_, W_, generator=g)
        wt = torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5
        assert_close(ops.conv_transpose2x2(xt.to(dev), wt.to(dev)), F.conv_transpose2d(xt, wt, stride=2), 1e-5,
                     f"conv transpose {(B_, Ci, Co, H_, W_)} (channel-group kernel for Cout % 14 == 0 or % 8 == 0)")
        # TransposeConvBlock (unet_block.py:296-299): statistics out of the transposed convolution's accumulators (odd Cout: three-pass norm)
        ref_t = F.leaky_relu(F.instance_norm(F.conv_transpose2d(xt.double(), wt.double(), stride=2), eps=1e-5), 0.2)
        assert_close(ops.conv_transpose2x2_instance_norm_act(xt.to(dev), wt.to(dev), 1e-5, ops.ACT_LEAKY, 0.2), ref_t, 1e-5,
                     f"conv transpose + instance norm + leaky {(B_, Ci, Co, H_, W_)}")
    xo = torch.randn(1, 14, 40, 40, generator=g) + 300.0           # a large mean: the tile-wise (mean, M2) merge keeps the variance
    wo = torch.randn(14, 14, 2, 2, generator=g).abs() / 14
    assert_close(ops.conv_transpose2x2_instance_norm_act(xo.to(dev), wo.to(dev), 1e-5, ops.ACT_LEAKY, 0.2),
                 F.leaky_relu(F.instance_norm(F.conv_transpose2d(xo.double(), wo.double(), stride=2), eps=1e-5), 0.2), 2e-4, "large-mean planes")
    y = torch.randn(2, 3, 17, 22, generator=g)
    assert_close(ops.concat_channels(x.to(dev), y.to(dev)), torch.cat([x, y], 1), 1e-12, "concat")


@pytest.mark.parametrize("shape", [(1, 64, 16, 32), (2, 16, 13, 37), (1, 24, 24, 72), (1, 20, 9, 8), (1, 64, 40, 372), (1, 32, 7, 100),
                                   (1, 64, 64, 30)])
@pytest.mark.parametrize("dil", [1, 2])
@pytest.mark.parametrize("pad_mode", ["zero", "replicate"])
def test_conv3x3_wino_vs_oracle(dev, shape, dil, pad_mode):
    """Plain 3x3 convolution into 64 channels on the Winograd kernel: both dilations, zero / replicate padding, bias, the three
    activations, widths with and without the 16-byte-aligned tile path, ragged tiles, channel counts off the 8-channel chunk."""
    from mridc_amd import ops
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(100 * dil + H + W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(64, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(64, generator=g)
    xp = F.pad(x, (dil, dil, dil, dil), mode="constant" if pad_mode == "zero" else "replicate")
    ref = F.conv2d(xp, w, b, dilation=dil)
    pm = ops.PAD_ZERO if pad_mode == "zero" else ops.PAD_REPLICATE
    assert ops.conv3x3_wino_supported(Cin, 64, 3, dil) and not ops.conv3x3_wino_supported(Cin, 48, 3, dil)
    if (H, W) == (13, 37) or (H, W) == (16, 32):            # 128 output channels: two 64-channel launches into one tensor (B = 2 too)
        w2 = torch.randn(128, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
        b2 = torch.randn(128, generator=g)
        assert_close(ops.conv2d(x.to(dev), w2.to(dev), b2.to(dev), dil, pm0 := (ops.PAD_ZERO if pad_mode == "zero" else ops.PAD_REPLICATE),
                                ops.ACT_RELU), F.relu(F.conv2d(xp, w2, b2, dilation=dil)), 1e-5, "128 output channels")
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    assert_close(ops.conv3x3_wino(xd, wd, bd, dil, pm), ref, 1e-5, f"wino conv {shape} dil {dil} {pad_mode}")
    assert_close(ops.conv3x3_wino(xd, wd, bd, dil, pm, ops.ACT_RELU), F.relu(ref), 1e-5, "relu")
    assert_close(ops.conv3x3_wino(xd, wd, None, dil, pm, ops.ACT_LEAKY, 0.25), F.leaky_relu(ref - b.view(1, -1, 1, 1), 0.25), 1e-5,
                 "leaky, no bias")
    # ops.conv2d routes these shapes here (Cin >= 16) and agrees with the direct kernels
    got = ops.conv2d(xd, wd, bd, dil, pm, ops.ACT_RELU)
    assert_close(got, F.relu(ref), 1e-5, "conv2d dispatch")
    old = ops.WINOGRAD_CONV
    ops.WINOGRAD_CONV = False
    try:
        direct = ops.conv2d(xd, wd, bd, dil, pm, ops.ACT_RELU)
    finally:
        ops.WINOGRAD_CONV = old
    assert_close(got, direct, 5e-6, "winograd vs direct kernel")
    # a changed weight (new version, same storage) is re-packed
    with torch.no_grad():
        wd.mul_(0.5)
    assert_close(ops.conv3x3_wino(xd, wd, bd, dil, pm), F.conv2d(xp, w * 0.5, b, dilation=dil), 1e-5, "re-pack after an in-place update")


@pytest.mark.parametrize("shape", [(1, 64, 16, 32), (2, 16, 13, 36), (1, 8, 9, 8), (1, 64, 40, 372), (1, 12, 5, 100), (1, 64, 8, 30)])
@pytest.mark.parametrize("pad_mode", ["zero", "replicate"])
def test_conv_to_complex_vs_oracle(dev, shape, pad_mode):
    """3x3 convolution into one complex image written as [B,H,W,2] (the tail of the CascadeNet / VSNet / RVN regularisers) on the
    RIM final-layer kernel: zero and replicate padding, every border case; W % 4 != 0 takes conv2d + permute."""
    from mridc_amd import ops
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(7 + H + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(2, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(2, generator=g)
    xp = F.pad(x, (1, 1, 1, 1), mode="constant" if pad_mode == "zero" else "replicate")
    pm = ops.PAD_ZERO if pad_mode == "zero" else ops.PAD_REPLICATE
    for bias in (b, None):
        ref = F.conv2d(xp, w, bias).permute(0, 2, 3, 1)
        got = ops.conv_to_complex(x.to(dev), w.to(dev), None if bias is None else bias.to(dev), 1, pm)
        assert got.is_contiguous()
        assert_close(got, ref, 1e-5, f"conv_to_complex {shape} {pad_mode} bias={bias is not None}")


@pytest.mark.parametrize("shape", [(2, 13, 18), (1, 33, 70), (1, 40, 372), (3, 1, 5)])
def test_conv1x1_64_vs_oracle(dev, shape):
    """1x1 convolution 64 -> 64 as a per-pixel GEMM: plain, with bias + each activation, and as the IndRNN cell (hh * h_prev + ReLU);
    ops.conv2d and ops.indrnn_cell route this shape to it."""
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(3 + H)
    x, hp = torch.randn(B, 64, H, W, generator=g), torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 1, 1, generator=g) / 8
    b, hh = torch.randn(64, generator=g), torch.randn(1, 64, 1, 1, generator=g)
    ref = F.conv2d(x, w, b)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    assert_close(ops.conv1x1_64(xd, wd, bd), ref, 1e-5, f"1x1 {shape}")
    assert_close(ops.conv2d(xd, wd, None, 1, ops.PAD_ZERO), F.conv2d(x, w), 1e-5, "conv2d dispatch, no bias")
    assert_close(ops.conv2d(xd, wd, bd, 1, ops.PAD_ZERO, ops.ACT_LEAKY, 0.1), F.leaky_relu(ref, 0.1), 1e-5, "leaky")
    assert_close(ops.indrnn_cell(xd, wd, bd, hh.to(dev), hp.to(dev), 1), oracle.rim.indrnn_cell(x, hp, w, b, hh, 1, 1), 1e-5, "IndRNN cell")
    assert_close(ops.indrnn_cell(xd, wd, None, hh.to(dev), None, 1), F.relu(F.conv2d(x, w)), 1e-5, "IndRNN cell, zero state")
    # 128 channels (the qCIRIM's cells): 2 x 2 blocks of the same GEMM
    x2, hp2 = torch.randn(B, 128, H, W, generator=g), torch.randn(B, 128, H, W, generator=g)
    w2 = torch.randn(128, 128, 1, 1, generator=g) / 11
    b2, hh2 = torch.randn(128, generator=g), torch.randn(1, 128, 1, 1, generator=g)
    assert ops.conv1x1_sq_supported(128, 128) and not ops.conv1x1_sq_supported(64, 128) and not ops.conv1x1_sq_supported(96, 96)
    assert_close(ops.conv2d(x2.to(dev), w2.to(dev), b2.to(dev), 1, ops.PAD_ZERO), F.conv2d(x2, w2, b2), 1e-5, "1x1 128 -> 128")
    assert_close(ops.indrnn_cell(x2.to(dev), w2.to(dev), b2.to(dev), hh2.to(dev), hp2.to(dev), 1),
                 oracle.rim.indrnn_cell(x2, hp2, w2, b2, hh2, 1, 1), 1e-5, "IndRNN cell, 128 features")


@pytest.mark.parametrize("shape", [(1, 640, 372), (2, 37, 75), (1, 19, 33), (3, 16, 32), (1, 5, 3), (1, 1, 1)])
def test_second_rim_layer_two_term_fp16_operands(shape, dev):
    """mrx_rim_layer2_f16 (the convolution's operands as two fp16 terms scaled by powers of two, three term products per multiply) against a
    float64 reference, the three-term bf16 kernel and the fp32 Winograd kernel: fp32-level error for any magnitude of x and w, with an exact
    and with a stale (1000 x too large) bound of max |x|; the tap products of the final convolution as well.  rim_block.py:233-246."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, H, W = shape
    F_ = 64
    g = torch.Generator().manual_seed(11 + sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    for xs, ws, with_state, with_bias in ((1.0, 1.0, True, True), (1e-4, 50.0, False, True), (2e4, 1e-3, True, False)):
        x, hp = r(B, F_, H, W).relu() * xs, r(B, F_, H, W).relu() * xs
        wc, wi, wf = r(F_, F_, 3, 3) / 24 * ws, r(F_, F_, 1, 1) / 8, r(2, F_, 3, 3) / 24
        b1, b2 = (r(F_) * 0.1 * xs * ws, r(F_) * 0.1) if with_bias else (None, None)
        hh = r(1, F_, 1, 1) * 0.5
        ref = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), None if b1 is None else b1.double(), dilation=2).relu()
        ref = Fn.conv2d(ref, wi.double(), None if b2 is None else b2.double())
        ref = Fn.relu(ref + hh.double() * hp.double() if with_state else ref)
        pk_h, pk_s, pk_w = ops.rim_layer2_f16_pack(wc, wi, wf), ops.rim_layer2_sb_pack(wc, wi, wf), ops.rim_layer_wino_pack(wc, wi)
        xmax = x.abs().max().reshape(1).contiguous()
        h = hp if with_state else None
        got = ops.rim_layer2_f16(x, pk_h, b1, b2, hh, h, xmax)
        stale = ops.rim_layer2_f16(x, pk_h, b1, b2, hh, h, xmax * 1000.0)
        sb = ops.rim_layer2_sb(x, pk_s, b1, b2, hh, h)
        win = ops.rim_layer_indrnn_wino(x, pk_w, F_, b1, b2, hh, h)
        e_h, e_st, e_w, e_sb = rel_l2(got, ref), rel_l2(stale, ref), rel_l2(win, ref), rel_l2(sb, ref)
        # (bias-dominated outputs: the 1x1 stage both split-operand kernels share accumulates on top of b_ih, the Winograd kernel adds it last)
        bound = max(2.5 * e_w, 1.05 * e_sb) + 5e-8
        assert e_h <= 6e-7 and e_h <= bound, (e_h, e_w, e_sb)
        assert e_st <= 6e-7 and e_st <= bound, (e_st, e_w, e_sb)
        assert rel_l2(got, sb) <= 8e-7
        h_t, taps = ops.rim_layer2_f16(x, pk_h, b1, b2, hh, h, xmax, want_taps=True)
        assert torch.equal(h_t, got)
        _, taps_sb = ops.rim_layer2_sb_taps(x, pk_s, b1, b2, hh, h)
        assert rel_l2(taps, taps_sb) <= 2e-6


def test_first_rim_layer_keeps_the_bound_of_its_outputs(dev):
    """mrx_rim_layer_indrnn_packed_xmax / _llg_xmax: the maximum of the outputs is folded into the device scalar with an atomic max (exact, never
    lowered), the outputs themselves are those of the plain calls."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    for B, H, W in ((1, 640, 372), (2, 37, 75), (1, 5, 3)):
        x, hp = r(B, 4, H, W), r(B, 64, H, W).relu()
        w, wi = r(64, 4, 5, 5) * 0.15, r(64, 64, 1, 1) * 0.2
        b, bi, hh = r(64) * 0.1, r(64) * 0.1, r(1, 64, 1, 1) * 0.5
        assert ops.rim_layer1_xmax_supported(4, 64, 5, 1)
        pk = ops.rim_layer_pack(w, wi)
        plain = ops.rim_layer_indrnn_packed(x, pk, 64, 5, 1, b, bi, hh, hp)
        xmax = torch.zeros(1, device=dev)
        got = ops.rim_layer_indrnn_packed(x, pk, 64, 5, 1, b, bi, hh, hp, xmax=xmax)
        assert torch.equal(got, plain)
        assert float(xmax) == float(plain.max())
        ops.rim_layer_indrnn_packed(x * 0.01, pk, 64, 5, 1, b, bi, hh, hp * 0.01, xmax=xmax)       # smaller outputs: the bound stays
        assert float(xmax) == float(plain.max())
        eta, part = r(B, H, W, 2), r(3, B, H, W, 2)
        a = ops.rim_layer_indrnn_packed_llg(eta, part, 3, 0.9, pk, 64, 5, 1, b, bi, hh, hp)
        xm2 = torch.zeros(1, device=dev)
        a2 = ops.rim_layer_indrnn_packed_llg(eta, part, 3, 0.9, pk, 64, 5, 1, b, bi, hh, hp, xmax=xm2)
        assert torch.equal(a, a2) and float(xm2) == float(a.max())


@pytest.mark.parametrize("shape", [(1, 128, 128, 256, 256, 2), (1, 64, 64, 256, 256, 1), (2, 48, 100, 37, 75, 2), (1, 32, 16, 19, 33, 1),
                                   (3, 40, 24, 8, 32, 2), (1, 130, 70, 5, 3, 2), (1, 33, 17, 1, 1, 1)])
def test_conv3x3_two_term_fp16_any_channels(shape, dev):
    """mrx_conv3x3_h (the U-Net's two-term fp16 kernel as a plain convolution: dilation 1 | 2, zero | replicate padding, bias + activation, one / two
    / four output-channel blocks per work item) against float64, and the route ops.conv2d takes for wide 3x3 layers (qRIM's 128 -> 128)."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    B, Cin, Cout, H, W, dil = shape
    g = torch.Generator().manual_seed(sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, w, b = r(B, Cin, H, W) * 3.0, r(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5, r(Cout) * 0.2
    assert ops.conv3x3_h_supported(Cin, Cout, 3, dil)
    for pad_mode, pad_name in ((ops.PAD_ZERO, "constant"), (ops.PAD_REPLICATE, "replicate")):
        xp = Fn.pad(x.double(), (dil, dil, dil, dil), mode=pad_name)
        for act, bias in ((ops.ACT_RELU, b), (ops.ACT_NONE, None), (ops.ACT_LEAKY, b)):
            ref = Fn.conv2d(xp, w.double(), None if bias is None else bias.double(), dilation=dil)
            ref = ref.relu() if act == ops.ACT_RELU else (Fn.leaky_relu(ref, 0.1) if act == ops.ACT_LEAKY else ref)
            got = ops.conv3x3_h(x, w, bias, dil, pad_mode, act, 0.1)
            assert rel_l2(got, ref) <= 1e-6, (shape, pad_name, act, rel_l2(got, ref))
    if Cin >= ops.H3X3_MIN_CIN and Cout >= 16 and not ops.conv3x3_sb_supported(Cin, Cout, 3, dil):
        seen = []
        orig = ops.conv3x3_h
        ops.conv3x3_h = lambda *a, **k: (seen.append(1), orig(*a, **k))[1]
        try:
            got = ops.conv2d(x, w, b, dil, ops.PAD_REPLICATE, ops.ACT_RELU)
        finally:
            ops.conv3x3_h = orig
        assert seen, "ops.conv2d did not take the two-term fp16 route"
        ref = Fn.conv2d(Fn.pad(x.double(), (dil, dil, dil, dil), mode="replicate"), w.double(), b.double(), dilation=dil).relu()
        assert rel_l2(got, ref) <= 1e-6


def test_cell_1x1_keeps_the_bound_its_consumer_needs(dev):
    """mrx_conv1x1_sq_xmax (C = 128): the outputs are those of mrx_conv1x1_sq, the device scalar holds exactly max |out|; ops.conv1x1_64 hands it
    to a following two-term fp16 convolution on the tensor (no mrx_max_abs launch), and a torch in-place write invalidates it."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(21)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(2, 128, 37, 75), r(2, 128, 37, 75).relu()
    w, b, hh = r(128, 128, 1, 1) / 11, r(128) * 0.1, r(1, 128, 1, 1) * 0.5
    keep = ops.H3X3_CONV
    try:
        ops.H3X3_CONV = False
        plain = ops.conv1x1_64(x, w, b, ops.ACT_RELU, 0.0, hh=hh, h_prev=hp)
        assert ops._lib.bound_of(plain) is None
        ops.H3X3_CONV = True
        got = ops.conv1x1_64(x, w, b, ops.ACT_RELU, 0.0, hh=hh, h_prev=hp)
    finally:
        ops.H3X3_CONV = keep
    assert torch.equal(got, plain)
    bound = ops._lib.bound_of(got)
    assert bound is not None and float(bound) == float(plain.abs().max())
    calls = []
    orig = ops.max_abs
    ops.max_abs = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        assert ops._plain_bound(got) is bound and not calls
        got.mul_(3.0)                                   # written by torch: the remembered bound no longer holds
        assert float(ops._plain_bound(got)) == float(got.abs().max()) and calls
    finally:
        ops.max_abs = orig


def test_a_library_write_into_a_bounded_tensor_drops_its_bound(dev):
    """The library writes through raw pointers and bumps no tensor version: every call drops the bounds kept for the ranges its non-const pointer
    arguments cover (_lib.written_pointer_args, read off the header) -- `out=` the tensor itself, `out=` a VIEW of it (another Python object:
    round 4's per-object attribute missed that one), and a tensor the caching allocator hands out again at the same address."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x = r(2, 64, 24, 40)
    w, b = r(64, 64, 3, 3) / 24, r(64) * 0.1

    def bounded():
        t = ops.conv3x3_sb(x, w, b, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0)
        assert ops._lib.bound_of(t) is not None
        return t
    t = bounded()
    ops.scale(x * 1e4, 1.0)                                               # a call that writes elsewhere leaves it alone
    assert ops._lib.bound_of(t) is not None
    big = r(2, 64, 24, 40) * 1e4
    ops.conv3x3_wino(big, w, b, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0, out=t)       # out = the tensor
    assert ops._lib.bound_of(t) is None
    t = bounded()
    ops.conv3x3_wino(big[:1], w, b, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0, out=t[1:])      # out = a view of its second half
    assert ops._lib.bound_of(t) is None
    y = ops.conv3x3_sb(t, w, b, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0)              # ... so the consumer takes the route that needs no bound
    import torch.nn.functional as Fn
    from tests._util import rel_l2
    assert rel_l2(y, Fn.conv2d(t.double(), w.double(), b.double(), padding=1)) <= 1e-6
    t = bounded()
    addr = t.data_ptr()
    del t                                                                        # freed: the next tensor of that size gets the address
    u = torch.empty(2, 64, 24, 40, device=dev)
    assert u.data_ptr() != addr or ops._lib.bound_of(u) is None


@pytest.mark.parametrize("dil,pad", [(1, "zero"), (2, "replicate"), (1, "replicate"), (2, "zero")])
def test_conv64_chain_keeps_bounds_and_switches_to_two_term_fp16(dev, dil, pad):
    """mrx_conv3x3_sb_chain: the first 64-channel convolution of a chain (no bound on its input: three-term bf16 operands) keeps max |y|, the
    next ones run on two-term fp16 operands scaled by it; every stage against float64 and against the bf16-only route (ops.SB_CHAIN = False)."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    g = torch.Generator().manual_seed(31 + dil)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    pm, pname = (ops.PAD_ZERO, "constant") if pad == "zero" else (ops.PAD_REPLICATE, "replicate")
    for B, H, W in ((1, 640, 372), (2, 37, 75), (1, 5, 3)):
        x = r(B, 64, H, W) * 40.0
        ws = [r(64, 64, 3, 3) / 24 for _ in range(3)]
        bs = [r(64) * 0.1, None, r(64) * 0.1]
        acts = [ops.ACT_RELU, ops.ACT_LEAKY, ops.ACT_NONE]
        calls = []
        orig = ops._lib.lib().mrx_conv3x3_sb_chain
        ref, got, plain = x.double(), x, x
        keep = ops.SB_CHAIN
        try:
            for i, (w, b, act) in enumerate(zip(ws, bs, acts)):
                ref = Fn.conv2d(Fn.pad(ref, (dil, dil, dil, dil), mode=pname), w.double(), None if b is None else b.double(), dilation=dil)
                ref = ref.relu() if act == ops.ACT_RELU else (Fn.leaky_relu(ref, 0.1) if act == ops.ACT_LEAKY else ref)
                ops.SB_CHAIN = True
                had_bound = ops._lib.bound_of(got) is not None
                got = ops.conv3x3_sb(got, w, b, dil, pm, act, 0.1)
                calls.append(had_bound)
                bound = ops._lib.bound_of(got)
                assert bound is not None and float(bound) == float(got.abs().max())
                ops.SB_CHAIN = False
                plain = ops.conv3x3_sb(plain, w, b, dil, pm, act, 0.1)
                assert ops._lib.bound_of(plain) is None
                assert rel_l2(got, ref) <= 8e-7 * (i + 1), (i, rel_l2(got, ref))
                assert rel_l2(got, plain) <= 1e-6 * (i + 1)
        finally:
            ops.SB_CHAIN = keep
        assert calls == [False, True, True]
        assert orig is not None


def test_gated_cell_keeps_the_bound_for_the_next_convolution(dev):
    """mrx_gated_cell_1x1_xmax: outputs of mrx_gated_cell_1x1, the scalar = max |out| exactly; the following 64-channel convolution takes the
    two-term fp16 form and agrees with float64."""
    import torch.nn.functional as Fn
    from mridc_amd import ops
    from tests._util import rel_l2
    g = torch.Generator().manual_seed(77)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    for gates in (3, 2):
        x, h = r(2, 64, 37, 75), r(2, 64, 37, 75)
        wi, wh, bi = r(gates * 64, 64, 1, 1) / 8, r(gates * 64, 64, 1, 1) / 8, r(gates * 64) * 0.1
        packed = ops.gated_cell_pack(wi, wh, gates)
        keep = ops.SB_CHAIN
        try:
            ops.SB_CHAIN = False
            plain = ops.gated_cell_1x1(x, h, packed, bi, gates)
            ops.SB_CHAIN = True
            got = ops.gated_cell_1x1(x, h, packed, bi, gates)
        finally:
            ops.SB_CHAIN = keep
        assert torch.equal(got, plain) and ops._lib.bound_of(plain) is None
        assert float(ops._lib.bound_of(got)) == float(plain.abs().max())
        w, b = r(64, 64, 3, 3) / 24, r(64) * 0.1
        y = ops.conv3x3_sb(got, w, b, 2, ops.PAD_REPLICATE, ops.ACT_RELU, 0.0)
        ref = Fn.conv2d(Fn.pad(plain.double(), (2, 2, 2, 2), mode="replicate"), w.double(), b.double(), dilation=2).relu()
        assert rel_l2(y, ref) <= 6e-7


def test_conv2dgru_cell_keeps_the_bound_of_its_activated_state(dev):
    """mrx_conv2dgru_cell_1x1_xmax: both outputs of mrx_conv2dgru_cell_1x1, the scalar = max ReLU(new state) exactly."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(78)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, h = r(2, 64, 37, 75).relu(), r(2, 64, 37, 75)
    wu, wr, wo = (r(64, 128, 1, 1) / 11 for _ in range(3))
    bias = r(3, 64) * 0.1
    packed = ops.conv2dgru_pack(wu, wr, wo)
    keep = ops.SB_CHAIN
    try:
        ops.SB_CHAIN = False
        new0, act0 = ops.conv2dgru_cell_1x1(x, h, packed, bias, True)
        ops.SB_CHAIN = True
        new1, act1 = ops.conv2dgru_cell_1x1(x, h, packed, bias, True)
    finally:
        ops.SB_CHAIN = keep
    assert torch.equal(new0, new1) and torch.equal(act0, act1) and ops._lib.bound_of(act0) is None
    assert float(ops._lib.bound_of(act1)) == float(act0.max())
