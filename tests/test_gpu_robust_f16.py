"""Two-term fp16 operands under heavy-tailed data (the default arithmetic of the matrix-core kernels, MRIDC_AMD_ARITH=f16x2).

x = (h1 + h2) 2^-k keeps 22 significant bits relative to a BLOCK scale 2^-k (one bound per tensor for the second RIM layer's convolution
input, per 32-pixel unit for the first layer's, per tile for the few-channel convolution, per pixel for every 1x1 contraction).  Elements far
below their block's maximum therefore lose relative -- not absolute -- precision: fp16 keeps the residual term normal down to 2^-17 of the
bound, below that the error relative to the ELEMENT grows by one bit per bit of distance.  Real knee data is heavy-tailed (a bright vessel
against near-zero background), so these tests judge the SMALL pixels element by element, not the tensor norm:

* one hot pixel 1e5 (and 1e6) times the plane's RMS: outputs away from it stay within 1e-5 of a float64 reference relative to their own size;
* a bound that went stale by 1e6 (a hidden state that shrank over the time-steps while the bound, an atomic max, never falls within a cascade:
  RIMBlock zeroes it once per call, i.e. per cascade): still 8e-6 on the whole tensor and 5e-5 element-wise (20 bits below the scale: one bit of relative precision lost per bit beyond 2^17);
* the reference has no such regime distinction (fp32 throughout: rim_block.py:217-249), hence the element-wise bar."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

F_ = 64


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _elementwise(got, ref, keep, rel, what):
    """|got - ref| <= rel * (|ref| + RMS of the kept outputs) on the kept elements (the outputs pass a ReLU: many are exactly zero, and an
    output that is small by cancellation carries the absolute error of its terms)."""
    g, r = got.double().cpu()[keep], ref.cpu()[keep]
    typical = float(r.pow(2).mean().sqrt()) + 1e-300
    bad = (g - r).abs() > rel * (r.abs() + typical)
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} small outputs off by more than {rel:g} of their own size " \
                                f"(worst {float(((g - r).abs() / (r.abs() + typical)).max()):.2e})"


def _layer2_ref(x, wc, bc, wi, bi, hh, hp):
    ref = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double(), dilation=2).relu()
    return Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())


@pytest.mark.parametrize("ratio,rel", [(1e5, 1e-5), (1e6, 3e-5)])
def test_second_layer_hot_pixel(dev, ratio, rel):
    """mrx_rim_layer2_f16: one pixel `ratio` times the rest in every channel; the bound is the true maximum (what layer 1 keeps)."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(3)
    H, W = 48, 96
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(1, F_, H, W).relu(), r(1, F_, H, W).relu()
    x[:, :, 20, 40] = ratio * (1.0 + x[:, :, 20, 40])
    wc, wi, wf = r(F_, F_, 3, 3) / 24, r(F_, F_, 1, 1) / 8, r(2, F_, 3, 3) / 24
    bc, bi, hh = r(F_) * 0.1, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    pk = ops.rim_layer2_f16_pack(wc, wi, wf)
    xmax = x.abs().max().reshape(1).contiguous()
    got = ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, xmax)
    ref = _layer2_ref(x, wc, bc, wi, bi, hh, hp)
    keep = torch.ones(1, F_, H, W, dtype=torch.bool)
    keep[:, :, 20 - 2:20 + 3, 40 - 2:40 + 3] = False          # outputs that see the hot pixel are judged by the norm below
    _elementwise(got, ref, keep, rel, f"layer 2, hot pixel x{ratio:g}")
    assert float((got.double() - ref).norm() / ref.norm()) <= 6e-7


def test_second_layer_stale_bound(dev):
    """The bound 1e6 times the data (a state that shrank by 1e6 since the bound was raised): every element 20 bits below the scale."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(4)
    H, W = 40, 70
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(1, F_, H, W).relu(), r(1, F_, H, W).relu()
    wc, wi = r(F_, F_, 3, 3) / 24, r(F_, F_, 1, 1) / 8
    bc, bi, hh = r(F_) * 0.1, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    pk = ops.rim_layer2_f16_pack(wc, wi, None)
    ref = _layer2_ref(x, wc, bc, wi, bi, hh, hp)
    exact = ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, x.abs().max().reshape(1).contiguous())
    if ops._lib.lib().mrx_checks_enabled():       # the CHECK build (-DMRX_CHECK_BOUNDS) refuses exactly this: a bound more than 2^16 above the data
        with pytest.raises(RuntimeError, match="more than 2\\^16 x max"):
            ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, (x.abs().max() * 1e6).reshape(1).contiguous())
        return
    stale = ops.rim_layer2_f16(x, pk, bc, bi, hh, hp, (x.abs().max() * 1e6).reshape(1).contiguous())
    e0, e1 = float((exact.double() - ref).norm() / ref.norm()), float((stale.double() - ref).norm() / ref.norm())
    assert e0 <= 6e-7 and e1 <= 8e-6, (e0, e1)
    _elementwise(stale, ref, torch.ones(1, F_, H, W, dtype=torch.bool), 5e-5, "layer 2, bound stale by 1e6")


def test_rim_block_zeroes_the_bound_every_cascade(dev):
    """The running bound lives for ONE RIMBlock.forward call (= one cascade): a huge state in one cascade does not set the scale of the next."""
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    torch.manual_seed(0)
    model = CIRIM(dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)).eval().to(dev)
    blk = model.cirim[0]
    d = {k: v.to(dev) for k, v in synthetic.make_slice(4, 32, 64, slice_idx=2).items()}
    seen = []
    import mridc_amd.ops as ops_mod
    names = ["rim_layer2_f16_cb8", "rim_layer2_f16_cb8_q"] if blk.cb8_states else ["rim_layer2_f16"]     # (whichever form of the layer the route takes)
    origs = {n: getattr(ops_mod, n) for n in names}

    def make_spy(orig):
        def spy(x, packed, b_conv, b_ih, hh, h_prev, xmax, **kw):
            seen.append(xmax)
            return orig(x, packed, b_conv, b_ih, hh, h_prev, xmax, **kw)
        return spy

    for n in names:
        setattr(ops_mod, n, make_spy(origs[n]))
    try:
        with torch.no_grad():
            blk(d["y"], d["y"], d["sensitivity_maps"], d["mask"])
            n1 = len(seen)
            blk(d["y"] * 1e-4, d["y"] * 1e-4, d["sensitivity_maps"], d["mask"])
    finally:
        for n in names:
            setattr(ops_mod, n, origs[n])
    assert n1 == blk.time_steps and len(seen) == 2 * n1
    assert all(t is seen[0] for t in seen[:n1]) and all(t is seen[n1] for t in seen[n1:]) and seen[n1] is not seen[0]   # one scalar per call
    first, second = float(seen[0]), float(seen[n1])                    # (read after both calls: the final bound of each)
    assert first > 0 and second > 0 and second < 1e-2 * first, (first, second)   # the second call's bound follows ITS (1e-4 x smaller) data


@pytest.mark.parametrize("ratio", [1e5])
def test_first_layer_hot_pixel(dev, ratio):
    """k_rim_layer1_sb<F16> (scale per 32-pixel unit from the unit's own 5 x 36 patch): a hot input pixel costs the units that see it relative
    precision on their small inputs only; every other unit is untouched."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(5)
    H, W = 40, 96
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp = r(1, 4, H, W), r(1, F_, H, W).relu()
    x[:, :, 17, 50] *= ratio
    wc, wi = r(F_, 4, 5, 5) / 10, r(F_, F_, 1, 1) / 8
    bc, bi, hh = r(F_) * 0.1, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    pk = ops.rim_layer_pack(wc, wi)
    got = ops.rim_layer_indrnn_packed(x, pk, F_, 5, 1, bc, bi, hh, hp)
    ref = Fn.conv2d(Fn.pad(x.double(), (2, 2, 2, 2), mode="replicate"), wc.double(), bc.double()).relu()
    ref = Fn.relu(Fn.conv2d(ref, wi.double(), bi.double()) + hh.double() * hp.double())
    far = torch.ones(1, F_, H, W, dtype=torch.bool)
    far[:, :, 17 - 2:17 + 3, 32:64 + 32] = False              # the units (row x 32 pixels, +-2 halo) whose patch holds the hot pixel
    far[:, :, 17 - 2:17 + 3, 50 - 34:50 + 35] = False
    _elementwise(got, ref, far, 1e-5, "layer 1, units that do not see the hot pixel")
    near = ~far
    near[:, :, 17 - 2:17 + 3, 50 - 2:50 + 3] = False           # (outputs that contain the hot pixel itself: norm criterion)
    _elementwise(got, ref, near, 1e-4, "layer 1, small inputs sharing a unit with the hot pixel")
    assert float((got.double() - ref).norm() / ref.norm()) <= 6e-7


def test_gated_cell_and_few_channel_conv_hot_pixel(dev):
    """k_gated_cell_sb<.., F16> scales per PIXEL, k_conv_sbs<.., F16> per tile: a hot pixel elsewhere leaves every other pixel / tile exact."""
    import oracle
    from mridc_amd import ops
    g = torch.Generator().manual_seed(6)
    H, W = 24, 80
    x, h = torch.randn(1, F_, H, W, generator=g), torch.randn(1, F_, H, W, generator=g)
    x[:, :, 10, 30] *= 1e5
    h[:, :, 5, 60] *= 1e5
    for gates, cell in ((3, oracle.rim.convgru_cell), (2, oracle.rim.convmgu_cell)):
        wi = torch.randn(gates * F_, F_, 1, 1, generator=g) / 8
        wh = torch.randn(gates * F_, F_, 1, 1, generator=g) / 8
        bi = torch.randn(gates * F_, generator=g)
        packed = ops.gated_cell_pack(wi.to(dev), wh.to(dev), gates)
        got = ops.gated_cell_1x1(x.to(dev), h.to(dev), packed, bi.to(dev), gates)
        ref = cell(x.double(), h.double(), wi.double(), bi.double(), wh.double(), 1, 1)
        keep = torch.ones(1, F_, H, W, dtype=torch.bool)
        keep[:, :, 10, 30] = False
        keep[:, :, 5, 60] = False
        _elementwise(got, ref, keep, 1e-5, f"gated cell ({gates} gates), pixels other than the hot ones")
    x8 = torch.randn(1, 8, H, W, generator=g)
    x8[:, :, 12, 40] *= 1e5
    w = torch.randn(128, 8, 5, 5, generator=g) / 14
    b = torch.randn(128, generator=g) * 0.1
    assert ops.conv_sbs_supported(8, 128, 5, 1)
    got = ops.conv_sbs(x8.to(dev), w.to(dev), b.to(dev), ops.PAD_REPLICATE, ops.ACT_RELU, 0.0)
    ref = Fn.conv2d(Fn.pad(x8.double(), (2, 2, 2, 2), mode="replicate"), w.double(), b.double()).relu()
    keep = torch.ones(1, 128, H, W, dtype=torch.bool)
    keep[:, :, 8 - 2:16 + 2, 32 - 2:64 + 2] = False            # the 8 x 32 tile holding the hot pixel (and its halo's reach)
    _elementwise(got, ref, keep, 1e-5, "few-channel convolution, tiles that do not see the hot pixel")
    assert float((got.double().cpu() - ref).norm() / ref.norm()) <= 6e-7
