"""The precision-16 inference route (csrc/rim_amp16.hip: mrx_amp16_layer1 / mrx_amp16_layer2, RIMBlock.precision = 16) -- the arithmetic the reference's own
inference configuration runs (`precision: 16`, projects/reconstruction/model_zoo/conf/base_cirim_run.yaml:132 = torch.autocast(float16) around forward).

Checked at three levels, all against the CPU oracle:
  * each layer kernel against float64 arithmetic on fp16-rounded operands (oracle.amp.fp16_kernel_arithmetic restates exactly that): what is left is the order of
    the fp32 sums and ONE fp16 rounding of the stored state (2^-11 relative per element = 2.8e-4 in rel-L2 for uniformly distributed mantissas);
  * a whole RIMBlock at 15 x 640 x 372 against the kernel-arithmetic oracle (tight) and against the oracle under torch.autocast(float16) -- the reference's
    semantics (autocast rounds every convolution's OUTPUT and keeps fp32 states; the kernels keep fp32 sums and fp16 states) -- on reference-init and on boosted
    weights, with the fp32 oracle beside them: a reduced-precision route cannot be closer to fp32 than autocast itself is;
  * SURVEY appendix C's stated bound for the fast mode (rel-L2 <= 3e-2, SSIM >= 0.99) on the 8-cascade chain, and the bounds frozen from the measurements.
The default route (precision 32) must be untouched by the switch: bit-identical results with and without the attribute."""
import os

import pytest
import torch

import oracle
from mridc_amd import synthetic
from tests._util import rel_l2

pytestmark = pytest.mark.gpu

F_ = 64
# (B, H, W): full tiles, ragged tiles in both directions, images smaller than a tile / than the halo, the headline plane
SHAPES = [(1, 96, 80), (2, 37, 45), (1, 5, 7), (1, 1, 1), (1, 33, 65), (1, 64, 372), (3, 16, 32)]
STATE_REL = 4e-4        # one fp16 rounding of the stored state (2^-11 per element) + fp32 summation order
FP32_REL = 2e-6         # fp32 results computed from the SAME fp16 operands: summation order only


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    return torch.device("cuda:0")


def _h16(x):
    return x.to(torch.float16).double()


def _c64(t):
    return t.detach().cpu().double()


def _weights(g, dev, cin=4):
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    w1, b1, wi1, bi1, hh1 = r(F_, cin, 5, 5) * 0.15, r(F_) * 0.1, r(F_, F_, 1, 1) * 0.2, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    w2, b2, wi2, bi2, hh2 = r(F_, F_, 3, 3) / 24, r(F_) * 0.1, r(F_, F_, 1, 1) / 8, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    wf, bf = r(2, F_, 3, 3) / 24, r(2) * 0.1
    return [t.to(dev) for t in (w1, b1, wi1, bi1, hh1, w2, b2, wi2, bi2, hh2, wf, bf)]


def _layer_ref(x, w, b, wi, bi, hh, hp, k, dil):
    """float64 layer on fp16-rounded operands: ReLU(conv_reppad(x16, w16) + b) -> fp16 -> ReLU(ih(g16, wi16) + bi + hh * hp)   (conv_layers.py:121-123,
    rnn_cells.py:384-391 through oracle.rim with oracle.amp's rounding)."""
    from oracle import rim as orim
    g = orim.conv_nonlinear(_h16(_c64(x)), _h16(_c64(w)), _c64(b), k, dil, "relu")
    return orim.indrnn_cell(_h16(g), _c64(hp) if hp is not None else torch.zeros_like(g), _h16(_c64(wi)), _c64(bi), _c64(hh), 1, 1)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_first_layer_against_float64_on_fp16_operands(dev, shape):
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(11 + H * W)
    w1, b1, wi1, bi1, hh1 = _weights(g, dev)[:5]
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x4, hp = r(B, 4, H, W), r(B, F_, H, W).relu()
    eta, part = r(B, H, W, 2), r(3, B, H, W, 2)
    pk = ops.amp16_layer1_pack(w1, wi1)
    hp16 = ops.amp16_from_nchw(hp)
    assert hp16.dtype == torch.float16 and tuple(hp16.shape) == (B, 4, H, W, 16)
    assert torch.equal(ops.amp16_to_nchw(hp16), hp.half().float())
    # x form, with and without a previous state
    for prev in (hp16, None):
        got = ops.amp16_to_nchw(ops.amp16_layer1(x4, None, None, 0, 1.0, pk, b1, bi1, hh1, prev))
        ref = _layer_ref(x4, w1, b1, wi1, bi1, hh1, None if prev is None else ops.amp16_to_nchw(prev), 5, 1)
        assert rel_l2(got, ref) <= STATE_REL, rel_l2(got, ref)
        assert float((got.cpu().double() - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max()) + 1e-6      # no element further than an fp16 ulp of the largest
    # (eta, partial planes) form = the x form on [eta, inv_sigma2 * sum of the planes] (rim_utils.py:61-67)
    sigma = 0.8
    for nparts in (1, 3):
        xx = torch.cat([eta.permute(0, 3, 1, 2), (part[:nparts].sum(0) / sigma ** 2).permute(0, 3, 1, 2)], 1).contiguous()
        got = ops.amp16_to_nchw(ops.amp16_layer1(None, eta, part, nparts, sigma, pk, b1, bi1, hh1, hp16))
        ref = _layer_ref(xx, w1, b1, wi1, bi1, hh1, hp.half().float(), 5, 1)
        assert rel_l2(got, ref) <= STATE_REL, (nparts, rel_l2(got, ref))
    # in place: the state overwritten by its successor gives the same bits
    a = ops.amp16_layer1(x4, None, None, 0, 1.0, pk, b1, bi1, hh1, hp16)
    buf = hp16.clone()
    b = ops.amp16_layer1(x4, None, None, 0, 1.0, pk, b1, bi1, hh1, buf, out=buf)
    assert b is buf and torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.amp16_layer1(x4, None, None, 0, 1.0, pk, b1, bi1, hh1, hp)          # an fp32 NCHW state is not what this route takes
    with pytest.raises(ValueError):
        ops.amp16_layer1(None, eta, part, 5, 1.0, pk, b1, bi1, hh1, hp16)       # at most four partial planes


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_second_layer_and_final_convolution_against_float64_on_fp16_operands(dev, shape):
    from mridc_amd import ops
    from oracle import rim as orim
    B, H, W = shape
    g = torch.Generator().manual_seed(23 + H * W)
    ws = _weights(g, dev)
    w2, b2, wi2, bi2, hh2, wf, bf = ws[5:]
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, hp, eta = r(B, F_, H, W).relu() * 3.0, r(B, F_, H, W).relu(), r(B, H, W, 2)
    x16, hp16 = ops.amp16_from_nchw(x), ops.amp16_from_nchw(hp)
    pk = ops.amp16_layer2_pack(w2, wi2, wf)
    for prev in (hp16, None):
        h, tq, te = ops.amp16_layer2(x16, pk, b2, bi2, hh2, prev)
        got = ops.amp16_to_nchw(h)
        ref = _layer_ref(x.half().float(), w2, b2, wi2, bi2, hh2, None if prev is None else hp.half().float(), 3, 2)
        assert rel_l2(got, ref) <= STATE_REL, rel_l2(got, ref)
        assert float((got.cpu().double() - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max()) + 1e-6
        # eta + final convolution (rim_block.py:239-248) from the kernel's OWN fp16 state: fp32 sums of exact fp16 products
        got_eta = ops.rim_final_gather_q(tq, te, bf, eta)
        ref_eta = _c64(eta) + (orim.conv_nonlinear(_c64(got), _h16(_c64(wf)), None, 3, 1, None) + _c64(bf).view(1, 2, 1, 1)).permute(0, 2, 3, 1)
        assert rel_l2(got_eta, ref_eta) <= FP32_REL, rel_l2(got_eta, ref_eta)
    # without tap planes: the same state
    h2 = ops.amp16_layer2(x16, pk, b2, bi2, hh2, hp16, want_taps=False)
    assert torch.equal(h2, ops.amp16_layer2(x16, pk, b2, bi2, hh2, hp16)[0])
    # in place
    buf = hp16.clone()
    h3, _, _ = ops.amp16_layer2(x16, pk, b2, bi2, hh2, buf, out=buf)
    assert h3 is buf and torch.equal(h3, h2)
    with pytest.raises(ValueError):
        ops.amp16_layer2(x, pk, b2, bi2, hh2, hp16)


def test_gather_inside_the_gradient_launch_takes_the_precision16_tap_planes(dev):
    """mrx_llg372_gather_q on the tap planes mrx_amp16_layer2 leaves = mrx_rim_final_gather_q on them, bit for bit (the same layout as the fp32-class route's)."""
    from mridc_amd import ops
    d = {k: v.to(dev) for k, v in synthetic.make_slice(6, 40, 372, slice_idx=1).items()}
    B, H, W = 1, 40, 372
    g = torch.Generator().manual_seed(5)
    ws = _weights(g, dev)
    w2, b2, wi2, bi2, hh2, wf, bf = ws[5:]
    x16 = ops.amp16_from_nchw(torch.randn(B, F_, H, W, generator=g).relu().to(dev))
    eta = torch.randn(B, H, W, 2, generator=g).to(dev)
    _, tq, te = ops.amp16_layer2(x16, ops.amp16_layer2_pack(w2, wi2, wf), b2, bi2, hh2, None)
    want = ops.rim_final_gather_q(tq, te, bf, eta)
    op = ops.llg372_prepare(ops.llg_prepare(d["y"], True, "ortho"), d["sensitivity_maps"], d["mask"], True, "ortho")
    _, nparts, got = ops.llg372_gather_q(eta, tq, te, bf, op, 1.0, "ortho")
    assert nparts >= 2 and torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.llg372_gather_q(eta, tq, te[:-1], bf, op, 1.0, "ortho")             # an undersized edge buffer is refused, not read past its end
    with pytest.raises(ValueError):
        ops.rim_final_gather_q(tq, te.double(), bf, eta)


def _cirim(cfg_over, scale, seed=0):
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, **cfg_over)
    torch.manual_seed(seed)
    model = CIRIM(cfg).eval()
    if scale != 1.0:
        with torch.no_grad():                          # the reference init is nearly linear: make the ReLUs bite
            for n, p in model.named_parameters():
                if n.endswith("rnn.ih.weight") or n.endswith("rnn.hh"):
                    p.mul_(scale)
    return cfg, model, {k: v.detach().clone() for k, v in model.state_dict().items()}


def _rim_cfg(cfg):
    return oracle.rim.RIMConfig(**{k: cfg[k] for k in ("recurrent_layer", "conv_filters", "conv_kernels", "conv_dilations", "conv_bias", "recurrent_filters",
                                                       "recurrent_kernels", "recurrent_dilations", "recurrent_bias", "depth", "no_dc", "fft_centered",
                                                       "fft_normalization", "spatial_dims", "coil_dim")}, time_steps=8)


# stated tolerances of one 8-step block at 15 x 640 x 372 (measured figures: profiles/r06_amp16_parity.txt), per weight set:
#   kernel_arithmetic: the kernels' own rounding points on the CPU; autocast_fp16: the reference's semantics; fp32: what precision 16 costs
BLOCK_TOL = {1.0: dict(kernel_arithmetic=2e-5, autocast_fp16=1e-4, fp32=1e-4), 5.0: dict(kernel_arithmetic=2e-4, autocast_fp16=2e-3, fp32=2e-3)}


# (torch's CPU fp16 convolutions take ~7 s per RIM step at 640 x 372 on the GPU box's host: ONE case at the headline size, the others on a quarter of the rows)
@pytest.mark.parametrize("scale,mask,H", [(1.0, "1d", 640), (5.0, "1d", 160), (1.0, "2d", 160), (5.0, "2d", 160)],
                         ids=["reference_init_1d_640", "x5_recurrent_weights_1d_160", "reference_init_2d_160", "x5_recurrent_weights_2d_160"])
def test_full_size_rim_block_precision16_against_the_three_oracles(dev, scale, mask, H):
    """One RIMBlock (8 steps, IndRNN 64) at 1 x 15 x 640 x 372 (and 15 x 160 x 372) with precision = 16: every estimate and both hidden states against the oracle
    in the kernels' arithmetic, under torch.autocast(float16) (the reference's `precision: 16`) and in fp32; 1-D column mask (the fused gradient + gather launch)
    and a 2-D mask (the general three-launch gradient, the gather as its own launch)."""
    cfg, model, sd = _cirim(dict(num_cascades=1), scale)
    d = synthetic.make_slice(15, H, 372, slice_idx=3)
    if mask == "2d":                                   # random 2-D points R ~ 4 with a fully sampled centre (stands in for the YAML's Poisson-2D)
        g = torch.Generator().manual_seed(3)
        m2 = torch.rand(1, 1, H, 372, 1, generator=g) < 0.22
        m2[:, :, :26, :15], m2[:, :, -26:, :15], m2[:, :, :26, -15:], m2[:, :, -26:, -15:] = True, True, True, True     # (non-centred k-space: DC at the corners)
        d = dict(d, mask=m2, y=d["kspace"] * m2)
    rc = _rim_cfg(cfg)
    p = {k[len("cirim.0."):]: v for k, v in sd.items() if k.startswith("cirim.0.")}
    refs = {}
    import contextlib
    for name, ctx in (("fp32", contextlib.nullcontext), ("autocast_fp16", oracle.amp.autocast_fp16), ("kernel_arithmetic", oracle.amp.fp16_kernel_arithmetic)):
        with ctx(), torch.no_grad():
            e, h = oracle.rim.rim_block_forward(p, rc, d["y"], d["y"], d["sensitivity_maps"], d["mask"], None, None, 1.0, False)
        refs[name] = (torch.stack([t.float() for t in e]), [t.float() for t in h])
    blk = model.cirim[0].to(dev)
    y, S, m = d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev)
    with torch.no_grad():
        e32, h32 = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
        blk.precision = 16
        try:
            assert blk._amp16_route()
            e16, h16 = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
        finally:
            blk.precision = None
        e32b, h32b = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
    for a, b in zip(list(e32) + list(h32), list(e32b) + list(h32b)):
        assert torch.equal(a, b), "the precision switch leaked into the default route"
    assert len(e16) == 8 and all(h.dtype == torch.float32 and tuple(h.shape) == (1, 64, H, 372) for h in h16)
    got = torch.stack(e16)
    tol = BLOCK_TOL[scale]
    meas = {}
    for name, (re_, rh_) in refs.items():
        meas[name] = (rel_l2(got, re_), rel_l2(got[-1], re_[-1]), rel_l2(h16[0], rh_[0]), rel_l2(h16[1], rh_[1]))
    print(f"[amp16 block scale {scale} mask {mask} H {H}] rel-L2 (all estimates, last estimate, h1, h2): " + ", ".join(f"{k} {v}" for k, v in meas.items())
          + f"; oracle autocast_fp16 vs fp32 {rel_l2(refs['autocast_fp16'][0], refs['fp32'][0]):.3g}; fp32 route vs fp32 oracle {rel_l2(torch.stack(e32), refs['fp32'][0]):.3g}")
    for name in tol:
        assert meas[name][0] <= tol[name] and meas[name][1] <= tol[name], (name, meas[name], tol[name])
    # the states: fp16 storage costs 2^-11 per element against every fp32-state oracle
    for name in tol:
        assert meas[name][2] <= 1e-3 and meas[name][3] <= 1e-3 + 10 * tol[name], (name, meas[name])


@pytest.mark.parametrize("shape", [(6, 37, 75), (4, 48, 40), (15, 24, 320)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("mask", ["1d", "2d"])
def test_rim_block_precision16_at_other_sizes(dev, shape, mask):
    """Away from W = 372 the gradient is its own [B,4,H,W] tensor (mrx_llg_hinv / mrx_llg) and layer 1 takes the x form; odd sizes, ragged tiles, two slices per call,
    states handed from one block call to the next (fp32 [B,64,H,W] at the API, fp16 inside): against the kernel-arithmetic oracle and under autocast."""
    C, H, W = shape
    cfg, model, sd = _cirim(dict(num_cascades=2), 3.0, seed=4)
    sl = [synthetic.make_slice(C, H, W, slice_idx=20 + i) for i in range(2)]
    y, S = torch.cat([s_["y"] for s_ in sl], 0), torch.cat([s_["sensitivity_maps"] for s_ in sl], 0)
    m = sl[0]["mask"]
    if mask == "2d":
        g = torch.Generator().manual_seed(9)
        m = torch.rand(1, 1, H, W, 1, generator=g) < 0.3
        y = torch.cat([s_["kspace"] for s_ in sl], 0) * m
    rc = _rim_cfg(cfg)
    p0 = {k[len("cirim.0."):]: v for k, v in sd.items() if k.startswith("cirim.0.")}
    p1 = {k[len("cirim.1."):]: v for k, v in sd.items() if k.startswith("cirim.1.")}
    refs = {}
    for name, ctx in (("autocast_fp16", oracle.amp.autocast_fp16), ("kernel_arithmetic", oracle.amp.fp16_kernel_arithmetic)):
        with ctx(), torch.no_grad():
            e0, h0 = oracle.rim.rim_block_forward(p0, rc, y, y, S, m, None, None, 1.0, False)
            e1, h1 = oracle.rim.rim_block_forward(p1, rc, e0, y, S, m, e0[-1], h0, 1.0, True)
        refs[name] = (torch.stack([t.float() for t in e0 + e1]), [t.float() for t in h1])
    b0, b1 = model.cirim[0].to(dev), model.cirim[1].to(dev)
    b0.precision = b1.precision = 16
    with torch.no_grad():
        g0, gh0 = b0(y.to(dev), y.to(dev), S.to(dev), m.to(dev), None, None, 1.0, keep_eta=False)
        g1, gh1 = b1(g0, y.to(dev), S.to(dev), m.to(dev), g0[-1], gh0, 1.0, keep_eta=True)
    got = torch.stack(list(g0) + list(g1))
    assert rel_l2(got, refs["kernel_arithmetic"][0]) <= 2e-4 and rel_l2(got, refs["autocast_fp16"][0]) <= 2e-3, (rel_l2(got, refs["kernel_arithmetic"][0]), rel_l2(got, refs["autocast_fp16"][0]))
    for j in range(2):
        assert gh1[j].dtype == torch.float32 and rel_l2(gh1[j], refs["kernel_arithmetic"][1][j]) <= 2e-3, rel_l2(gh1[j], refs["kernel_arithmetic"][1][j])


def test_eight_cascades_precision16_final_image(dev):
    """All 8 cascades x 8 time-steps (BASELINE.json's headline model) at 1 x 15 x 128 x 372 with `precision: 16` handed in through the trainer, as the reference
    does: final image against the oracle under torch.autocast(float16) and in fp32 -- SURVEY appendix C's bound for the fast mode (rel-L2 <= 3e-2, SSIM >= 0.99)
    and the bound frozen from the measurement."""
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    cfg = dict(synthetic.CIRIM_BASELINE_CFG)

    class Trainer:              # what the reference's model receives: an object with the configured `precision`
        precision = 16
    torch.manual_seed(0)
    model = CIRIM(cfg, trainer=Trainer()).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert all(b.precision == 16 and b._amp16_route() for b in model.cirim)
    d = synthetic.make_slice(15, 128, 372, slice_idx=2)
    with torch.no_grad():
        ref32 = oracle.models.cirim_forward(sd, cfg, d["y"], d["sensitivity_maps"], d["mask"], None, d["target"])[-1][-1]
        with oracle.amp.autocast_fp16():
            ref16 = oracle.models.cirim_forward(sd, cfg, d["y"], d["sensitivity_maps"], d["mask"], None, d["target"])[-1][-1]
        model = model.to(dev)
        out = next(model(d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev), None, d["target"].to(dev)))[-1][-1]
    a = lambda t: torch.view_as_real(t.detach().cpu().to(torch.complex64)) if t.is_complex() else t.detach().cpu()  # noqa: E731
    r16, r32 = rel_l2(a(out), a(ref16)), rel_l2(a(out), a(ref32))
    img = lambda t: (a(t).pow(2).sum(-1).sqrt() if a(t).shape[-1] == 2 else a(t).abs())  # noqa: E731
    norm = lambda t: (img(t) / img(t).max()).reshape(1, 1, *img(t).shape[-2:])  # noqa: E731
    ssim = 1.0 - float(oracle.metrics.ssim_loss(norm(out), norm(ref16), torch.tensor([1.0])))           # losses/ssim.py:46-61 on abs / max images
    print(f"[amp16 chain] rel-L2 vs autocast_fp16 {r16:.3g}, vs fp32 {r32:.3g}, oracle autocast vs fp32 {rel_l2(a(ref16), a(ref32)):.3g}, SSIM {ssim}")
    assert r16 <= 3e-2 and r32 <= 3e-2                     # SURVEY appendix C, the stated bound of the fast mode
    assert r16 <= 2e-4 and r32 <= 2e-4                     # frozen from the measurement (reference-init weights)
    assert ssim >= 0.99
