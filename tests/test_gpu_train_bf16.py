"""BASELINE config 4 with bf16 storage (csrc/train_bf16.hip, conv_bf16.hip OUT 1 / 2): every kernel of the tape against a float64 restatement of the
reference's autocast arithmetic (base_cirim_train.yaml:180: convolutions on half-precision operands returning half-precision tensors; hidden
states, hh * hx, eta and the loss fp32) -- rim_block.py:230-246, conv_layers.py:121-123, rnn_cells.py:384-391 and their autograd."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests._util import rel_l2

pytestmark = pytest.mark.gpu
bf = oracle.amp.bf16_round


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _flips(got, ref, ulps=1.01):
    """Fraction of elements further than one bf16 ulp from the reference (a bf16 result of an fp32 sum may round the other way)."""
    g, r = got.double().cpu(), ref.double()
    tol = ulps * 2.0 ** -8 * r.abs().clamp_min(1e-30) + 1e-30
    return float(((g - r).abs() > tol).double().mean())


def test_pair_tensors_are_bf16_round_to_nearest_even(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 5, 7, generator=g) * torch.logspace(-3, 3, 6).view(1, 6, 1, 1)
    p = ops.f32_to_pairs(x.to(dev))
    assert p.dtype == torch.int32 and tuple(p.shape) == (2, 3, 5, 7)
    assert torch.equal(ops.pairs_to_f32(p).cpu(), bf(x))
    lo = (p.cpu() & 0xffff).to(torch.int32)
    assert torch.equal((lo << 16).view(torch.float32), bf(x)[:, 0::2])


def _blk(t):
    """NCHW -> channel-blocked [B,C/8,H,W,8] (plain torch: independent of ops.cb8_from_nchw)."""
    return None if t is None else t.view(t.shape[0], t.shape[1] // 8, 8, t.shape[2], t.shape[3]).permute(0, 1, 3, 4, 2).contiguous()


def _mask_words(h):
    """(h > 0) of [B,64,H,W] as int32 [B,H,W,2]: word w, bit 16 c2 + 4 k + m = channel 32 c2 + 8 k + 4 w + m (include/mridc_amd.h, mrx_tl_layer_fwd)."""
    B, _, H, W = h.shape
    out = torch.zeros(B, H, W, 2, dtype=torch.int64)
    for c in range(64):
        c2, k, w, m = c >> 5, (c >> 3) & 3, (c >> 2) & 1, c & 3
        out[..., w] |= (h[:, c] > 0).to(torch.int64) << (16 * c2 + 4 * k + m)
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out)
    return out.to(torch.int32)


def _unblk(t):
    return t.permute(0, 1, 4, 2, 3).reshape(t.shape[0], t.shape[1] * 8, t.shape[2], t.shape[3])


@pytest.mark.parametrize("shape", [(1, 4, 5, 19, 45), (2, 64, 3, 21, 40), (1, 4, 5, 8, 32)], ids=lambda s: f"B{s[0]}_cin{s[1]}_k{s[2]}_{s[3]}x{s[4]}")
@pytest.mark.parametrize("with_state", [False, True])
def test_training_layer_forward_rounds_where_autocast_rounds(dev, shape, with_state):
    from mridc_amd import ops
    B, Cin, k, H, W = shape
    dil = 1 if k == 5 else 2
    g = torch.Generator().manual_seed(B * 100 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    cw, cb = torch.randn(64, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, torch.randn(64, generator=g) * 0.1
    w_ih, b_ih, hh = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1, torch.randn(1, 64, 1, 1, generator=g) * 0.5
    w_fin = torch.randn(2, 64, 3, 3, generator=g) / 24 if Cin == 64 else None
    hp = torch.randn(B, 64, H, W, generator=g) if with_state else None
    pad = dil * (k - 1) // 2
    xp = F.pad(bf(x).double(), (pad,) * 4, mode="replicate")
    a = F.relu(bf((F.conv2d(xp, bf(cw).double(), None, dilation=dil) + bf(cb).double().view(1, -1, 1, 1)).float()))
    u = bf((F.conv2d(a.double(), bf(w_ih).double()) + bf(b_ih).double().view(1, -1, 1, 1)).float())
    h = F.relu(u + (hh * hp if with_state else 0.0))
    d = lambda t: None if t is None else t.to(dev)  # noqa: E731
    # hidden states are channel-blocked [B,8,H,W,8] in this tape: h, h_prev, and the 64-channel layer's input
    a_p, h_cb, taps, hm = ops.tl_layer_fwd(_blk(d(x)) if Cin == 64 else d(x), d(cw), d(cb), d(w_ih), d(b_ih), d(hh), _blk(d(hp)), d(w_fin), want_mask=True)
    assert tuple(h_cb.shape) == (B, 8, H, W, 8)
    h_g = _unblk(h_cb)
    assert torch.equal(hm.cpu(), _mask_words(h_g.cpu()))              # (h > 0) as 64 bits per pixel: all the cell's backward reads of h
    a_g = ops.pairs_to_f32(a_p)
    assert _flips(a_g, a) <= 2e-3 and rel_l2(a_g, a) <= 2e-3, (_flips(a_g, a), rel_l2(a_g, a))
    assert rel_l2(h_g, h) <= 3e-3, rel_l2(h_g, h)
    # given ITS OWN a, the cell stage is exact up to the order of the fp32 sums
    u2 = bf((F.conv2d(a_g.cpu().double(), bf(w_ih).double()) + bf(b_ih).double().view(1, -1, 1, 1)).float())
    h2 = F.relu(u2 + (hh * hp if with_state else 0.0))
    assert _flips(h_g, h2, 1.0) <= 2e-3 and rel_l2(h_g, h2) <= 1e-3
    if w_fin is not None:
        eta = torch.randn(B, H, W, 2, generator=g)
        hb = F.pad(bf(h_g.cpu()).double(), (1,) * 4, mode="replicate")
        want = eta + bf(F.conv2d(hb, bf(w_fin).double()).float()).permute(0, 2, 3, 1)
        got = ops.tl_final_gather(taps, eta.to(dev))
        assert rel_l2(got, want) <= 1e-3, rel_l2(got, want)
    else:
        assert taps is None


@pytest.mark.parametrize("variant", ["full", "no_carry", "first_step", "many_tiles", "many_tiles_no_carry", "many_tiles_first_step"])
def test_cell_backward_in_one_pass(dev, variant):
    """mrx_tl_cell_bwd + mrx_tl_cell_reduce against the written-out backward of rnn_cells.py:390 / conv_layers.py:123 under autocast
    (many_tiles: more tiles than workgroups -- the persistent loop and its per-workgroup accumulators)."""
    from mridc_amd import ops
    B, H, W = (2, 19, 45) if not variant.startswith("many_tiles") else (1, 640, 372)
    g = torch.Generator().manual_seed(11)
    dh = bf(torch.randn(B, 64, H, W, generator=g))
    dH = torch.randn(B, 64, H, W, generator=g) if not variant.endswith("no_carry") else None
    h = F.relu(torch.randn(B, 64, H, W, generator=g))
    hp = torch.randn(B, 64, H, W, generator=g) if not variant.endswith("first_step") else None
    a = bf(F.relu(torch.randn(B, 64, H, W, generator=g)))
    w_ih, hh = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(1, 64, 1, 1, generator=g) * 0.5
    up = dh + (dH if dH is not None else 0.0)
    gq = torch.where(h > 0, up, torch.zeros(()))
    gb = bf(gq)
    da = bf(F.conv_transpose2d(gb.double(), bf(w_ih).double()).float())
    ga = torch.where(a > 0, da, torch.zeros(()))
    want = dict(dw=torch.einsum("bohw,bihw->oi", gb.double(), a.double()), dbih=gb.double().sum((0, 2, 3)),
                dhh=(gq.double() * hp.double()).sum((0, 2, 3)) if hp is not None else torch.zeros(64, dtype=torch.float64), db=ga.double().sum((0, 2, 3)))
    d = lambda t: None if t is None else t.to(dev)  # noqa: E731
    blk = _blk
    part = ops.tl_cell_part(B, H, W, dev)
    part.fill_(float("nan"))                               # `first` must overwrite the slots
    dhp, ga_p = ops.tl_cell_bwd(ops.f32_to_pairs(d(dh)), d(blk(dH)), d(blk(h)), d(blk(hp)), ops.f32_to_pairs(d(a)), d(w_ih), None, d(hh), part, True)
    ga_g = ops.pairs_to_f32(ga_p)
    assert _flips(ga_g, ga) <= 2e-3 and rel_l2(ga_g, ga) <= 2e-3, (_flips(ga_g, ga), rel_l2(ga_g, ga))
    if hp is not None:
        assert tuple(dhp.shape) == (B, 8, H, W, 8)             # channel-blocked, as the next time-step's call reads it
        assert rel_l2(_unblk(dhp), gq * hh) <= 1e-6
    else:
        assert dhp is None
    # second time-step (adds), with the state given as its mask words: the same gradients bit for bit
    part_h = part.clone()
    dhp_m, ga_m = ops.tl_cell_bwd(ops.f32_to_pairs(d(dh)), d(blk(dH)), d(_mask_words(h)), d(blk(hp)), ops.f32_to_pairs(d(a)), d(w_ih), None, d(hh), part, False)
    assert torch.equal(ga_m, ga_p) and (dhp is None or torch.equal(dhp_m, dhp))
    ops.tl_cell_bwd(ops.f32_to_pairs(d(dh)), d(blk(dH)), d(blk(h)), d(blk(hp)), ops.f32_to_pairs(d(a)), d(w_ih), None, d(hh), part_h, False)
    assert torch.equal(part_h, part)
    dw, dbih, dhh, db = (torch.full(s, 1.0, device=dev) for s in ((64, 64, 1, 1), (64,), (64,), (64,)))
    ops.tl_cell_reduce(part, B, H, W, dw, dbih, dhh, db)
    assert rel_l2(dw.reshape(64, 64).cpu().double() - 1.0, 2 * want["dw"]) <= 1e-5
    assert rel_l2(dbih.cpu().double() - 1.0, 2 * want["dbih"]) <= 1e-5
    if hp is not None:
        assert rel_l2(dhh.cpu().double() - 1.0, 2 * want["dhh"]) <= 1e-5
    else:
        assert float((dhh - 1.0).abs().max()) == 0.0
    assert rel_l2(db.cpu().double() - 1.0, 2 * want["db"]) <= 2e-3        # sums of the kernel's own bf16 ga (rounding flips against the reference's)
    assert rel_l2(db.cpu().double() - 1.0, 2 * ga_g.cpu().double().sum((0, 2, 3))) <= 1e-5


@pytest.mark.parametrize("case", [(64, 64, 3, 2, True, True), (64, 2, 3, 1, False, True), (4, 64, 5, 1, True, False)], ids=lambda c: f"{c[0]}to{c[1]}_k{c[2]}d{c[3]}")
def test_data_gradient_with_bf16_results(dev, case):
    """mrx_tl_dgrad + mrx_tl_fold_edges: the data gradient of a replicate-padded convolution on bf16 operands, rounded to bf16 (pair tensor or
    fp32 holding bf16 values), against float64 autograd."""
    from mridc_amd import ops
    Cin, Cout, k, dil, pairs_in, pairs_out = case
    B, H, W = 2, 19, 45
    g = torch.Generator().manual_seed(Cin + Cout)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    pad = dil * (k - 1) // 2
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.pad(x, (pad,) * 4, mode="replicate"), bf(w).double(), dilation=dil).backward(dy.double())
    want = x.grad
    dyg = ops.f32_to_pairs(dy.to(dev)) if pairs_in else dy.to(dev)
    got = ops.tl_dgrad(dyg, w.to(dev), dil, pairs_out)
    got = ops.pairs_to_f32(got) if pairs_out else got
    assert tuple(got.shape) == (B, Cin, H, W)
    assert torch.equal(bf(got.cpu()), got.cpu())                     # bf16 values
    assert rel_l2(got, want) <= 3e-3 and _flips(got, bf(want.float()), 2.0) <= 5e-3, (rel_l2(got, want), _flips(got, bf(want.float()), 2.0))


@pytest.mark.parametrize("case", [(64, 3, 2), (4, 5, 1), (5, 5, 1), (1, 5, 1)], ids=lambda c: f"cin{c[0]}_k{c[1]}d{c[2]}")
def test_weight_gradient_from_a_pair_tensor(dev, case):
    """mrx_conv_wgrad_bf16_pairs = mrx_conv_wgrad_bf16_any on the same bf16 values: bit-identical, and against float64."""
    from mridc_amd import ops
    Cin, k, dil = case
    B, H, W = 1, 21, 44
    g = torch.Generator().manual_seed(Cin)
    x, dy = torch.randn(B, Cin, H, W, generator=g), bf(torch.randn(B, 64, H, W, generator=g))
    dyp = ops.f32_to_pairs(dy.to(dev))
    keep = ops.TL_WGRAD_IN
    try:
        ops.TL_WGRAD_IN = False
        got = ops.conv_wgrad_bf16_pairs(x.to(dev), dyp, k, dil, ops.PAD_REPLICATE)
        same = ops.conv_wgrad_bf16(x.to(dev), dy.to(dev), k, dil, ops.PAD_REPLICATE)
        assert torch.equal(got, same)                         # the generic kernels: the same bf16 values, the same order of sums
    finally:
        ops.TL_WGRAD_IN = keep
    got = ops.conv_wgrad_bf16_pairs(x.to(dev), dyp, k, dil, ops.PAD_REPLICATE)      # (5x5 on <= 5 channels: mrx_tl_wgrad_in)
    pad = dil * (k - 1) // 2
    w = torch.zeros(64, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.pad(bf(x).double(), (pad,) * 4, mode="replicate"), w, dilation=dil).backward(dy.double())
    assert rel_l2(got, w.grad) <= 1e-5
    acc = torch.ones_like(got)
    ops.conv_wgrad_bf16_pairs(x.to(dev), dyp, k, dil, ops.PAD_REPLICATE, out=acc, accumulate=True)
    assert rel_l2(acc.cpu().double() - 1.0, w.grad) <= 1e-5
    if Cin == 64:                                             # the tape's form: x channel-blocked -- the same values in the same order
        assert torch.equal(ops.conv_wgrad_bf16_pairs(_blk(x.to(dev)), dyp, k, dil, ops.PAD_REPLICATE), got)


@pytest.mark.parametrize("shape", [(1, 21, 44), (2, 19, 45), (1, 640, 372)], ids=lambda s: "x".join(map(str, s)))
def test_final_convolution_weight_gradient_from_blocked_states(dev, shape):
    """mrx_conv_wgrad_bf16_xcb = mrx_conv_wgrad_bf16_any (3x3, 64 -> 2) on the channel-blocked hidden state: bit-identical."""
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H)
    x, dy = torch.randn(B, 64, H, W, generator=g).to(dev), torch.randn(B, 2, H, W, generator=g).to(dev)
    want = ops.conv_wgrad_bf16(x, dy, 3, 1, ops.PAD_REPLICATE)
    assert torch.equal(ops.conv_wgrad_bf16_xcb(_blk(x), dy, ops.PAD_REPLICATE), want)
    acc = torch.ones_like(want)
    ops.conv_wgrad_bf16_xcb(_blk(x), dy, ops.PAD_REPLICATE, out=acc, accumulate=True)
    assert rel_l2(acc - 1.0, want) <= 1e-6


@pytest.mark.parametrize("seed,boost", [(0, 1.0), (5, 3.0)])
def test_bf16_storage_tape_against_the_three_oracle_arithmetics(dev, seed, boost):
    """The whole tape (training.cirim_forward_backward, 'bf16') on a 2-cascade CIRIM at 4 x 48 x 40 (plumbing of the cascades: slices of the flat
    gradient, per-cascade partial sums; the quantitative check is the one-cascade test below and the headline-size test) against oracle/amp.py: the kernels' own arithmetic
    (operand AND result rounding restated on the CPU: tight), torch.autocast (the reference's semantics: differs by autocast's bf16 accumulation of
    weight gradients over the time-steps) and the fp32 oracle (loose: what bf16 costs)."""
    from mridc_amd import autograd as ag
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    from tests._util import detie_l1_target
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
    torch.manual_seed(seed)
    model = CIRIM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if boost != 1.0 and (n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh")):
                p_.mul_(boost)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s = synthetic.make_slice(4, 48, 40, slice_idx=2)
    with torch.no_grad():
        s["target"] = detie_l1_target(s["target"], oracle.models.cirim_forward(state, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"]), 3e-3)
    refs = {m: oracle.amp.cirim_loss_and_gradients(state, cfg, s, mode, round_results=rr)
            for m, mode, rr in (("kernels", "bf16_operands", True), ("autocast", "autocast_bf16", False), ("fp32", "fp32", False))}
    model = model.to(dev).train()
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    assert training._tape_supported(model, batch) and all(training._tl_supported(b) for b in model.cirim)
    ag.set_precision("bf16")
    try:
        loss = training.cirim_forward_backward(model, batch, "bf16")
    finally:
        ag.set_precision("f32")
    names = [n for n, _ in model.named_parameters() if not n.endswith("dc_weight")]
    got = torch.cat([dict(model.named_parameters())[n].grad.detach().cpu().reshape(-1).double() for n in names])
    err = {}
    for m, (ref_loss, grads) in refs.items():
        want = torch.cat([grads[n].reshape(-1).double() for n in names])
        err[m] = (float((got - want).norm() / want.norm()), abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    print("bf16-storage tape vs oracle arithmetics (whole gradient, loss):", err)
    # (two cascades on a 48 x 40 image: the second cascade starts from an estimate that already differs by bf16 rounding flips, and 1920 pixels do not
    # average the flipped ReLU masks / l1 signs out -- one cascade on this data agrees to 1.5e-3, tools/probe/train_parity.py)
    assert err["kernels"][0] <= 1e-1 and err["kernels"][1] <= 2e-4, err
    assert err["autocast"][0] <= 1e-1 and err["autocast"][1] <= 2e-3, err
    assert err["fp32"][0] <= 1.5e-1 and err["fp32"][1] <= 2e-2, err


@pytest.mark.parametrize("seed,boost", [(0, 1.0), (5, 3.0)])
def test_bf16_storage_tape_one_cascade_is_the_kernel_arithmetic(dev, seed, boost):
    """One cascade at 4 x 48 x 40: the tape against the CPU restatement of its own arithmetic (bf16 operands, bf16 convolution results, fp32 sums)."""
    from mridc_amd import autograd as ag
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)
    torch.manual_seed(seed)
    model = CIRIM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if boost != 1.0 and (n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh")):
                p_.mul_(boost)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s = synthetic.make_slice(4, 48, 40, slice_idx=0)
    ref_loss, ref = oracle.amp.cirim_loss_and_gradients(state, cfg, s, "bf16_operands", round_results=True)
    model = model.to(dev).train()
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    ag.set_precision("bf16")
    try:
        loss = training.cirim_forward_backward(model, batch, "bf16")
    finally:
        ag.set_precision("f32")
    names = [n for n, _ in model.named_parameters() if not n.endswith("dc_weight")]
    got = torch.cat([dict(model.named_parameters())[n].grad.detach().cpu().reshape(-1).double() for n in names])
    want = torch.cat([ref[n].reshape(-1).double() for n in names])
    e = float((got - want).norm() / want.norm())
    print(f"one cascade, seed {seed}: whole gradient vs the kernels' arithmetic {e:.3e}")
    assert e <= 1.5e-2 and abs(float(loss) - float(ref_loss)) <= 2e-4 * abs(float(ref_loss)), (e, float(loss), float(ref_loss))


@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 640, 372)], ids=lambda s: "x".join(map(str, s)))
def test_fused_loss_path_kernels_equal_the_separate_operators(dev, shape):
    """The tape's glue launches folded together (round 4): gather + per-workgroup maxima (mrx_tl_final_gather_max), the l1 term from those partial maxima
    (mrx_absl1_loss_mp), loss backward + eta_grad_in (mrx_absl1_loss_bwd_eta), eta_grad_out from the adjoint's partial planes (mrx_eta_grad_out_parts):
    each against the separate operators it replaces -- bit for bit where the order of additions is kept, to rounding for the loss sums."""
    from mridc_amd import _lib, ops
    L = _lib.lib()
    B, H, W = shape
    plane = H * W
    g = torch.Generator().manual_seed(H)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    taps, eta, tgt = r(B, 18, H, W) * 0.1, r(B, H, W, 2), torch.rand(B, H, W, generator=g).to(dev)
    st = _lib.stream_ptr()
    # forward
    want_eta = ops.tl_final_gather(taps, eta)
    want_m = ops.max_abs(want_eta, complex_modulus=True).reshape(1)
    want_out2, work = torch.empty(2, device=dev), torch.empty(int(L.mrx_absl1_work_floats()), device=dev)
    _lib.check(L.mrx_absl1_loss(_lib.ptr(want_eta), _lib.ptr(tgt), _lib.ptr(want_m), _lib.ptr(want_out2), _lib.ptr(work), B * plane, st), "loss")
    nmp = int(L.mrx_tl_final_gather_max_count(B, H, W))
    got_eta, mp, got_m, got_out2 = torch.empty_like(eta), torch.empty(nmp, device=dev), torch.empty(1, device=dev), torch.empty(2, device=dev)
    _lib.check(L.mrx_tl_final_gather_max(_lib.ptr(taps), _lib.ptr(eta), _lib.ptr(got_eta), _lib.ptr(mp), B, H, W, st), "gather_max")
    _lib.check(L.mrx_absl1_loss_mp(_lib.ptr(got_eta), _lib.ptr(tgt), _lib.ptr(mp), nmp, _lib.ptr(got_m), _lib.ptr(got_out2), _lib.ptr(work), B * plane, st), "loss_mp")
    assert torch.equal(got_eta, want_eta) and torch.equal(got_m, want_m)
    assert torch.equal(got_out2, want_out2)                       # (same partial sums, same final order)
    # backward
    wdev, carry = torch.full((1,), 0.37, device=dev), r(B, H, W, 2)
    for cr in (carry, None):
        gl, want_tot, want_d2 = torch.empty_like(eta), torch.empty_like(eta), torch.empty(B, 2, H, W, device=dev)
        _lib.check(L.mrx_absl1_loss_bwd(_lib.ptr(want_eta), _lib.ptr(tgt), _lib.ptr(want_m), _lib.ptr(want_out2), _lib.ptr(wdev), 1.0, _lib.ptr(gl), B * plane, st), "bwd")
        _lib.check(L.mrx_eta_grad_in(_lib.ptr(cr), _lib.ptr(gl), _lib.ptr(want_tot), _lib.ptr(want_d2), B, plane, st), "eta_in")
        got_tot, got_d2 = torch.empty_like(eta), torch.empty(B, 2, H, W, device=dev)
        _lib.check(L.mrx_absl1_loss_bwd_eta(_lib.ptr(want_eta), _lib.ptr(tgt), _lib.ptr(want_m), _lib.ptr(want_out2), _lib.ptr(wdev), 1.0, _lib.ptr(cr),
                                            _lib.ptr(got_tot), _lib.ptr(got_d2), B, plane, st), "bwd_eta")
        assert torch.equal(got_tot, want_tot) and torch.equal(got_d2, want_d2)
    # eta_grad_out from partial planes: t4's gradient channels = post * (sum of the planes in order)
    n_p, post = 4, 0.64
    parts, g4, tot = r(n_p, B, H, W, 2), r(B, 4, H, W), r(B, H, W, 2)
    s = parts[0].clone()
    for k in range(1, n_p):
        s = s + parts[k]
    t4 = torch.cat([g4[:, :2], (s * post).permute(0, 3, 1, 2)], 1).contiguous()
    want, got = torch.empty_like(tot), torch.empty_like(tot)
    _lib.check(L.mrx_eta_grad_out(_lib.ptr(tot), _lib.ptr(g4), _lib.ptr(t4), _lib.ptr(want), B, plane, st), "eta_out")
    _lib.check(L.mrx_eta_grad_out_parts(_lib.ptr(tot), _lib.ptr(g4), _lib.ptr(parts), n_p, post, _lib.ptr(got), B, plane, st), "eta_out_parts")
    assert rel_l2(got, want) <= 1e-7                              # (s * post may contract into the addition: one rounding apart)


@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 9, 32), (1, 640, 372)], ids=lambda s: "x".join(map(str, s)))
def test_data_gradient_64_with_weights_in_lds_is_bit_identical(dev, shape):
    """mrx_tl_dgrad's 3x3 dilation-2 64 -> 64 kernel (weights resident in LDS, persistent workgroups: k_tl_dgrad64) against the generic kernel that
    streams them from L2 (mrx_tl_dgrad_l2w): the same steps in the same order -- interior (pairs) and the folded edges bit for bit."""
    from mridc_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    dy = ops.f32_to_pairs(bf(torch.randn(B, 64, H, W, generator=g)).to(dev))
    got = ops.tl_dgrad(dy, w, 2, True)
    want = ops.tl_dgrad(dy, w, 2, True, weights_in_lds=False)
    assert got.dtype == torch.int32 and torch.equal(got, want)
    # the first layer's 5x5 64 -> Cin <= 4 gradient (compact weight table: the four real rows of every fragment)
    for cin in (4, 2):
        w5 = (torch.randn(64, cin, 5, 5, generator=g) / 40).to(dev)
        got5 = ops.tl_dgrad(dy, w5, 1, False)
        want5 = ops.tl_dgrad(dy, w5, 1, False, weights_in_lds=False)
        assert got5.dtype == torch.float32 and tuple(got5.shape) == (B, cin, H, W) and torch.equal(got5, want5)
