"""Channel-blocked hidden states h[b][c / 8][y][x][c % 8] of the two RIM layer kernels (mrx_cb8_convert, mrx_rim_layer1_cb8,
mrx_rim_layer2_f16_cb8): the layout between the kernels of a time-step is free (rim_block.py:230-246 only hands states from step to step), the
arithmetic is that of the NCHW entry points, so every result must be BIT-identical to theirs -- operation by operation and for a whole cascade."""
import pytest
import torch

pytestmark = pytest.mark.gpu

F_ = 64
SHAPES = [(1, 640, 372), (2, 37, 75), (1, 19, 33), (3, 16, 32), (1, 5, 3), (1, 1, 1)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_layout_round_trip(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(1)
    for B, C, H, W in ((1, 64, 640, 372), (2, 16, 7, 13), (3, 8, 1, 1)):
        x = torch.randn(B, C, H, W, generator=g).to(dev)
        y = ops.cb8_from_nchw(x)
        assert tuple(y.shape) == (B, C // 8, H, W, 8)
        assert torch.equal(y, x.view(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous())
        assert torch.equal(ops.cb8_to_nchw(y), x)
    with pytest.raises(RuntimeError):
        ops.cb8_from_nchw(torch.zeros(1, 12, 4, 4, device=dev))      # channels must come in blocks of eight


@pytest.mark.parametrize("shape", SHAPES)
def test_both_layers_bit_identical_to_the_nchw_kernels(shape, dev):
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(3 + sum(shape))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x4, hp = r(B, 4, H, W), r(B, F_, H, W).relu()
    eta, part = r(B, H, W, 2), r(3, B, H, W, 2)
    w1, wi1 = r(F_, 4, 5, 5) / 10, r(F_, F_, 1, 1) / 8
    w2, wi2, wf = r(F_, F_, 3, 3) / 24, r(F_, F_, 1, 1) / 8, r(2, F_, 3, 3) / 24
    bc, bi, hh = r(F_) * 0.1, r(F_) * 0.1, r(1, F_, 1, 1) * 0.5
    pk1, pk2 = ops.rim_layer_pack(w1, wi1), ops.rim_layer2_f16_pack(w2, wi2, wf)
    hpc = ops.cb8_from_nchw(hp)
    for state, statec in ((hp, hpc), (None, None)):
        xm, xmc = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        ref1 = ops.rim_layer_indrnn_packed(x4, pk1, F_, 5, 1, bc, bi, hh, state, xmax=xm)
        got1 = ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, statec, xmc)
        assert torch.equal(ops.cb8_to_nchw(got1), ref1) and torch.equal(xm, xmc)
        ref1l = ops.rim_layer_indrnn_packed_llg(eta, part, 3, 0.9, pk1, F_, 5, 1, bc, bi, hh, state, xmax=xm)
        got1l = ops.rim_layer1_cb8(None, eta, part, 3, 0.9, pk1, bc, bi, hh, statec, xmc)
        assert torch.equal(ops.cb8_to_nchw(got1l), ref1l) and torch.equal(xm, xmc)
        ref2, taps = ops.rim_layer2_f16(ref1, pk2, bc, bi, hh, state, xm, want_taps=True)
        got2, tapsc = ops.rim_layer2_f16_cb8(got1, pk2, bc, bi, hh, statec, xmc, want_taps=True)
        assert torch.equal(ops.cb8_to_nchw(got2), ref2) and torch.equal(tapsc, taps)
        assert torch.equal(ops.cb8_to_nchw(ops.rim_layer2_f16_cb8(got1, pk2, bc, bi, hh, statec, xmc)), ref2)     # without the tap stage
    # in place: the new state over the old one
    xm = torch.zeros(1, device=dev)
    ref1 = ops.rim_layer_indrnn_packed(x4, pk1, F_, 5, 1, bc, bi, hh, hp, xmax=xm)
    ref2 = ops.rim_layer2_f16(ref1, pk2, bc, bi, hh, hp, xm)
    s1, s2 = hpc.clone(), hpc.clone()
    assert ops.rim_layer1_cb8(x4, None, None, 0, 1.0, pk1, bc, bi, hh, s1, xm, out=s1) is s1
    ops.rim_layer2_f16_cb8(s1, pk2, bc, bi, hh, s2, xm, out=s2)
    assert torch.equal(ops.cb8_to_nchw(s1), ref1) and torch.equal(ops.cb8_to_nchw(s2), ref2)


@pytest.mark.parametrize("mask_kind", ["1d", "2d"])
def test_cascade_on_channel_blocked_states_is_bit_identical(dev, mask_kind):
    """RIMBlock.forward with cb8_states on and off: the same estimates at every time-step and the same states handed back (converted), with
    states passed in from a previous call as well."""
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    torch.manual_seed(0)
    model = CIRIM(dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)).eval().to(dev)
    blk = model.cirim[0]
    assert blk._f16_route() and blk.cb8_states
    d = {k: v.to(dev) for k, v in synthetic.make_slice(4, 40, 372 if mask_kind == "2d" else 64, slice_idx=1).items()}
    mask = d["mask"]
    if mask_kind == "2d":
        gen = torch.Generator().manual_seed(7)
        mask = (torch.rand(1, 1, 40, 372, 1, generator=gen) < 0.3).to(dev)
    y = d["y"] * mask
    from mridc_amd import ops
    with torch.no_grad():
        eq, hq = blk(y, y, d["sensitivity_maps"], mask)             # default: channel-blocked states AND the tap products pre-summed along x (ops.RIM_TAPS_Q)
        try:
            ops.RIM_TAPS_Q = False                                  # eighteen tap planes: the arithmetic of the NCHW route, addition by addition
            e1, h1 = blk(y, y, d["sensitivity_maps"], mask)
            e1b, h1b = blk(y, y, d["sensitivity_maps"], mask, hx=h1)
            blk.cb8_states = False
            e0, h0 = blk(y, y, d["sensitivity_maps"], mask)
            e0b, h0b = blk(y, y, d["sensitivity_maps"], mask, hx=h0)
        finally:
            ops.RIM_TAPS_Q = True
            del blk.cb8_states
        _, none = blk(y, y, d["sensitivity_maps"], mask, _want_hx=False)
    assert none is None
    for a, b in zip(e1 + e1b + list(h1) + list(h1b), e0 + e0b + list(h0) + list(h0b)):
        assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(eq + list(hq), e0 + list(h0)):                  # the pre-summed form: the order of nine additions per estimate
        assert a.shape == b.shape and float((a.double() - b.double()).norm() / b.double().norm()) <= 2e-6


def test_cascade_with_the_tap_gather_folded_into_the_gradient_is_bit_identical(dev):
    """RIMBlock.forward at W = 372 with a column mask: the nine-tap gather that ends a time-step folded into the next step's gradient launch
    (mrx_llg372_gather; default) against the separate launches -- every estimate and the states bit for bit."""
    from mridc_amd import ops, synthetic
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    torch.manual_seed(0)
    model = CIRIM(dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)).eval().to(dev)
    blk = model.cirim[0]
    d = {k: v.to(dev) for k, v in synthetic.make_slice(6, 24, 372, slice_idx=2).items()}
    assert ops.LLG372_GATHER and ops.LLG372_NO_Y and ops.RIM_TAPS_Q
    with torch.no_grad():
        eq, hq = blk(d["y"], d["y"], d["sensitivity_maps"], d["mask"])          # default: the gather folded in AND the taps pre-summed along x
        try:
            ops.RIM_TAPS_Q = False
            e1, h1 = blk(d["y"], d["y"], d["sensitivity_maps"], d["mask"])      # the gather folded in, eighteen tap planes
            ops.LLG372_GATHER = False
            e0, h0 = blk(d["y"], d["y"], d["sensitivity_maps"], d["mask"])      # separate launches
        finally:
            ops.LLG372_GATHER, ops.RIM_TAPS_Q = True, True
    assert len(e1) == len(e0) == len(eq) == blk.time_steps
    for a, b in zip(e1 + list(h1), e0 + list(h0)):
        assert a.shape == b.shape and torch.equal(a, b)
    # the pre-summed form adds the same nine products in another order: round-off of one fp32 sum per estimate, carried through eight time-steps
    for a, b in zip(eq + list(hq), e0 + list(h0)):
        assert a.shape == b.shape and float((a.double() - b.double()).norm() / b.double().norm()) <= 2e-6


@pytest.mark.parametrize("shape", [(1, 96, 80), (2, 37, 45), (1, 5, 7), (1, 64, 372), (1, 33, 65), (1, 20, 64)], ids=lambda s: "x".join(map(str, s)))
def test_tap_products_pre_summed_along_x(dev, shape):
    """mrx_rim_layer2_f16_cb8_q + mrx_rim_final_gather_q against the eighteen-plane route and float64: the state is bit-identical, the estimate differs by the order
    of nine additions; tile borders (every 32nd column), image borders (replicate padding) and ragged last tiles; at W = 372 the gather inside the gradient launch
    (mrx_llg372_gather_q) is bit-identical to mrx_rim_final_gather_q."""
    from mridc_amd import ops
    B, H, W = shape
    F = 64
    g = torch.Generator().manual_seed(H * W + 1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    w2, wi2, wf, bf = r(F, F, 3, 3) / 24, r(F, F, 1, 1) / 8, r(2, F, 3, 3) / 24, r(2) * 0.1
    bc, bi, hh = r(F) * 0.1, r(F) * 0.1, r(1, F, 1, 1) * 0.5
    x, hp, eta = r(B, F, H, W).relu() * 3.0, r(B, F, H, W).relu(), r(B, H, W, 2)
    xm = x.abs().max().reshape(1).contiguous()
    # the reference of this operator test is the CPU oracle in float64 (oracle.rim: conv_layers.py:121-123, rnn_cells.py:384-391, rim_block.py:239-248), not a device library
    from oracle import rim as orim
    c64 = lambda t: t.detach().cpu().double()  # noqa: E731
    gd = orim.conv_nonlinear(c64(x), c64(w2), c64(bc), 3, 2, "relu")
    ref = orim.indrnn_cell(gd, c64(hp), c64(wi2), c64(bi), c64(hh), 1, 1)
    ref_eta = (c64(eta) + (orim.conv_nonlinear(ref, c64(wf), None, 3, 1, None) + c64(bf).view(1, 2, 1, 1)).permute(0, 2, 3, 1)).to(dev)
    pk = ops.rim_layer2_f16_pack(w2, wi2, wf)
    xc, hc = ops.cb8_from_nchw(x), ops.cb8_from_nchw(hp)
    d_h, d_t = ops.rim_layer2_f16_cb8(xc, pk, bc, bi, hh, hc, xm, want_taps=True)
    d_eta = ops.rim_final_gather(d_t, bf, eta)
    q_h, q_t, q_e = ops.rim_layer2_f16_cb8_q(xc, pk, bc, bi, hh, hc, xm)
    q_eta = ops.rim_final_gather_q(q_t, q_e, bf, eta)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    assert torch.equal(q_h, d_h)
    assert rel(q_eta, ref_eta) <= 5e-7 and rel(d_eta, ref_eta) <= 5e-7, (rel(q_eta, ref_eta), rel(d_eta, ref_eta))
    assert float((q_eta - d_eta).abs().max()) <= 4e-6 * float(d_eta.abs().max())
    # the pre-summed planes are what they claim to be: Q[dy, co](x) = P[dy, 0](x - 1) + P[dy, 1](x) + P[dy, 2](x + 1) inside a tile
    P = d_t.double().reshape(B, 3, 3, 2, H, W)
    left, right = torch.roll(P[:, :, 0], 1, -1), torch.roll(P[:, :, 2], -1, -1)
    left[..., 0], right[..., -1] = P[:, :, 0][..., 0], P[:, :, 2][..., -1]      # replicate padding at the image border
    cols = torch.arange(W)
    left[..., (cols % 32 == 0) & (cols > 0)] = 0.0                               # ... and the neighbouring tile's share left to `edges`
    right[..., (cols % 32 == 31) & (cols < W - 1)] = 0.0
    want_q = (left + P[:, :, 1] + right).reshape(B, 6, H, W)
    assert rel(q_t.reshape(B, 3, H, W, 2).permute(0, 1, 4, 2, 3).reshape(B, 6, H, W), want_q) <= 2e-7       # taps_q is [B,3,H,W,2]: a pair per kernel row


def test_gradient_launch_gather_on_pre_summed_taps_is_bit_identical_to_the_gather_kernel(dev):
    from mridc_amd import ops, synthetic
    d = {k: v.to(dev) for k, v in synthetic.make_slice(6, 40, 372, slice_idx=1).items()}
    B, C, H, W = 1, 6, 40, 372
    g = torch.Generator().manual_seed(7)
    tq = torch.randn(B, 3, H, W, 2, generator=g).to(dev)
    te = torch.randn(int(ops._lib.lib().mrx_rim_taps_q_edge_floats(B, H, W)), generator=g).to(dev)
    eta, bf = torch.randn(B, H, W, 2, generator=g).to(dev), torch.randn(2, generator=g).to(dev)
    want = ops.rim_final_gather_q(tq, te, bf, eta)
    op = ops.llg372_prepare(ops.llg_prepare(d["y"], True, "ortho"), d["sensitivity_maps"], d["mask"], True, "ortho")
    _, nparts, got = ops.llg372_gather_q(eta, tq, te, bf, op, 1.0, "ortho")
    assert nparts >= 2 and torch.equal(got, want)
