"""GPU parity of the training-path backward kernels against torch autograd of the same fp32 operations on
the CPU (the reference trains through torch autograd: rim_block.py:217-249).  Tolerance: rel-L2 <= 1e-5 per operator."""
import pytest
import torch
import torch.nn.functional as F

from tests._util import assert_close, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref_conv(x, w, b, k, dil, pad_mode):
    p = dil * (k - 1) // 2
    xp = F.pad(x, (p, p, p, p), mode="replicate" if pad_mode == "replicate" else "constant")
    return F.conv2d(xp, w, b, dilation=dil)


CASES = [  # B, Cin, Cout, H, W, k, dil
    (1, 4, 64, 16, 32, 5, 1), (2, 64, 64, 13, 37, 3, 2), (1, 64, 64, 9, 70, 1, 1), (1, 64, 2, 12, 40, 3, 1), (1, 16, 64, 8, 33, 3, 1),
    (1, 24, 4, 7, 9, 3, 1), (1, 64, 64, 40, 372, 3, 2),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("pad_mode", ["replicate", "zero"])
def test_conv_backward_vs_autograd(dev, case, pad_mode):
    from mridc_amd import ops
    B, Cin, Cout, H, W, k, dil = case
    g = torch.Generator().manual_seed(H * W + Cin + k)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)).requires_grad_(True)
    dy = torch.randn(B, Cout, H, W, generator=g)
    _ref_conv(x, w, None, k, dil, pad_mode).backward(dy)
    pm = ops.PAD_REPLICATE if pad_mode == "replicate" else ops.PAD_ZERO
    dw = ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm)
    assert_close(dw, w.grad, 1e-5, f"weight gradient {case} {pad_mode}")
    acc = torch.full_like(dw, 0.5)
    ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm, out=acc, accumulate=True)
    assert_close(acc, w.grad + 0.5, 1e-5, "accumulating form")
    dx = ops.conv_dgrad(dy.to(dev), w.detach().to(dev), dil, pm)
    assert_close(dx, x.grad, 1e-5, f"data gradient {case} {pad_mode}")
    # determinism: the reductions have a fixed order
    assert torch.equal(dw, ops.conv_wgrad(x.detach().to(dev), dy.to(dev), k, dil, pm))


@pytest.mark.parametrize("shape", [(2, 64, 13, 18), (1, 5, 9, 370), (1, 64, 40, 372)])
def test_relu_and_indrnn_backward_vs_autograd(dev, shape):
    from mridc_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B + C + H)
    pre = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    bias = torch.randn(C, generator=g, requires_grad=True)
    hh = torch.randn(1, C, 1, 1, generator=g, requires_grad=True)
    hp = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    dy = torch.randn(B, C, H, W, generator=g)
    y = F.relu(pre + bias.view(1, -1, 1, 1) + hh * hp)       # the IndRNN cell's output stage (rnn_cells.py:390)
    y.backward(dy)
    dpre, dhp, sums = ops.relu_bwd(dy.to(dev), y.detach().to(dev), hp.detach().to(dev), hh.detach().to(dev))
    assert_close(dpre, pre.grad, 1e-6, "dpre")
    assert_close(dhp, hp.grad, 1e-6, "dh_prev")
    assert_close(sums[:, 0], bias.grad, 1e-5, "bias gradient")
    assert_close(sums[:, 1], hh.grad.reshape(-1), 1e-5, "hh gradient")
    dpre2, none, sums2 = ops.relu_bwd(dy.to(dev), y.detach().to(dev))
    assert none is None and torch.equal(dpre2, dpre) and torch.equal(sums2[:, 0], sums[:, 0])


def _small_cirim(dev, cascades=2):
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    cfg = dict(synthetic.CIRIM_BASELINE_CFG)
    cfg["num_cascades"] = cascades
    torch.manual_seed(3)
    model = CIRIM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0, 0.05)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s = synthetic.make_slice(3, 24, 20, slice_idx=1)
    # tie-free data: no l1 term of any estimate within 1e-3 of zero, no two pixels tying for max |eta| (tests/_util.py: a sign on a
    # round-off boundary would make the gradient comparison a coin toss)
    import oracle
    from tests._util import detie_l1_target
    with torch.no_grad():
        pred = oracle.models.cirim_forward(state, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
    s["target"] = detie_l1_target(s["target"], pred)
    return cfg, model.to(dev), state, s


def test_cirim_training_gradients_vs_oracle_autograd(dev):
    """Loss and every parameter gradient of a 2-cascade x 8-step CIRIM (IndRNN, no_dc) on the HIP training path against torch
    autograd of the CPU oracle with the reference's loss (cirim.py:199-247)."""
    import oracle
    from mridc_amd import training
    cfg, model, state, s = _small_cirim(dev)
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    pred = oracle.models.cirim_forward(p, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
    T_ = oracle.models.cirim_time_steps(cfg["time_steps"])
    ref_loss = oracle.models.cirim_process_loss(s["target"], pred, torch.nn.L1Loss(), T_, cfg["num_cascades"])
    ref_loss.backward()
    model.train()
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    etas = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
    loss = training.cirim_l1_loss(etas, batch["target"], model.time_steps, len(model.cirim))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach())), (float(loss.detach()), float(ref_loss.detach()))
    checked = 0
    for name, prm in model.named_parameters():
        if name.endswith("dc_weight"):
            continue                                  # unused by no_dc cascades: no gradient on either side
        ref = p[name].grad
        assert ref is not None and prm.grad is not None, name
        assert_close(prm.grad, ref, 2e-3, f"gradient of {name}")
        checked += 1
    assert checked == 2 * 11
    # the inference path is untouched by train(): same outputs in eval mode under no_grad
    model.eval()
    with torch.no_grad():
        out = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
    assert_close(torch.view_as_real(out[-1][-1]), torch.view_as_real(etas[-1][-1].detach()), 1e-4, "train vs eval forward")


def test_training_step_matches_torch_adam(dev):
    """FlatParameters + AdamFlat: two training steps move the parameters exactly as torch.optim.Adam does on the same gradients."""
    from mridc_amd import training
    cfg, model, state, s = _small_cirim(dev, cascades=1)
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    flat = training.FlatParameters(model)
    opt = training.AdamFlat(flat, lr=1e-3, betas=(0.9, 0.98))
    shadow = [p.detach().clone().requires_grad_(True) for p in flat.params]
    ref_opt = torch.optim.Adam(shadow, lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    losses = []
    for _ in range(2):
        losses.append(float(training.training_step(model, flat, opt, batch)))
        for sp, p in zip(shadow, flat.params):
            sp.grad = p.grad.detach().clone()
        ref_opt.step()
        for sp, p in zip(shadow, flat.params):
            assert_close(p.data, sp.data, 1e-6, "parameter after the Adam step")
    assert losses[1] < losses[0], losses
    assert flat.flat.numel() == sum(p.numel() for p in model.parameters())
    # the updated weights are the ones the kernels use next (packed-weight caches are keyed on the parameter versions the
    # optimizer bumps): the eval forward equals the oracle's on the current state_dict
    import oracle
    model.eval()
    cur = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        out = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
    ref = oracle.models.cirim_forward(cur, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
    assert_close(torch.view_as_real(out[-1][-1]), torch.view_as_real(ref[-1][-1]), 1e-4, "forward after two optimizer steps")
    assert any(not torch.equal(cur[k], state[k]) for k in state), "the optimizer moved the parameters"


@pytest.mark.parametrize("mask_kind", ["1d", "2d"])
def test_llg_backward_is_its_own_adjoint(dev, mask_kind):
    """log_likelihood_gradient's backward (the forward kernel on the incoming gradient with y = 0) against torch autograd of the
    oracle, for the one-launch row-invariant form and the general three-launch form."""
    import oracle
    from mridc_amd import autograd as ag
    from mridc_amd import ops
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 1, 3, 20, 24
    eta = torch.randn(B, H, W, 2, generator=g, requires_grad=True)
    y = torch.randn(B, C, H, W, 2, generator=g)
    S = torch.randn(B, C, H, W, 2, generator=g) / C ** 0.5
    mask = (torch.rand(1, 1, 1, W, 1, generator=g) < 0.4) if mask_kind == "1d" else (torch.rand(1, 1, H, W, 1, generator=g) < 0.4)
    y = y * mask
    dout = torch.randn(B, 4, H, W, generator=g)
    for centered, norm in ((False, "backward"), (True, "ortho")):
        eta.grad = None
        ref = oracle.rim.log_likelihood_gradient(eta, y, S, mask, 1.3, centered, norm, [-2, -1], 1)
        ref.backward(dout)
        e = eta.detach().to(dev).requires_grad_(True)
        hinv = mask_kind == "1d"
        data = ops.llg_prepare(y.to(dev), centered, norm) if hinv else y.to(dev)
        out = ag.LogLikelihoodGradient.apply(e, data, S.to(dev), mask.to(dev), 1.3, centered, norm, hinv)
        assert_close(out, ref.detach(), 1e-5, f"llg forward {mask_kind} {norm}")
        out.backward(dout.to(dev))
        assert_close(e.grad, eta.grad, 1e-5, f"llg backward {mask_kind} {norm}")


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_explicit_tape_matches_autograd_tape(dev, precision):
    """training.cirim_forward_backward (the written-out backward: accumulation inside the kernels, one cascade alive at a time) against the
    torch-autograd tape over the same kernels: same loss, same gradients up to the order of the fp32 additions.  (bf16: the fp32-storage form of
    the explicit tape -- the autograd tape's kernels; the bf16-storage tape has its own tests in test_gpu_train_bf16.py.)"""
    from mridc_amd import autograd as ag
    from mridc_amd import training
    cfg, model, state, s = _small_cirim(dev)
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    assert training._tape_supported(model, batch)
    ag.set_precision(precision)
    keep_storage = training.BF16_STORAGE
    training.BF16_STORAGE = False
    try:
        model.train()
        etas = next(model(batch["y"], batch["sensitivity_maps"], batch["mask"], None, batch["target"]))
        loss = training.cirim_l1_loss(etas, batch["target"], model.time_steps, len(model.cirim))
        loss.backward()
        want = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        for p in model.parameters():
            p.grad = None
        done = []
        got_loss = training.cirim_forward_backward(model, batch, precision, done.append)
        assert done == list(range(len(model.cirim)))
        assert abs(float(got_loss) - float(loss.detach())) <= 2e-6 * abs(float(loss.detach()))
        for n, g in want.items():
            assert_close(dict(model.named_parameters())[n].grad, g, 2e-5, f"{precision} tape gradient of {n}")
        # through training_step: FlatParameters slices per cascade, Adam on the accumulated buffer
        flat = training.FlatParameters(model)
        assert len(flat.cascade_slices) == len(model.cirim) and flat.cascade_slices[0][0] == 1      # dc_weight comes first
        assert sum(b - a for a, b in flat.cascade_slices) + sum(b - a for a, b in flat.rest_slices) == flat.numel
        import copy
        model_b = copy.deepcopy(model)
        flat_b = training.FlatParameters(model_b)
        opt = training.AdamFlat(flat, lr=1e-4, betas=(0.9, 0.98))
        opt_b = training.AdamFlat(flat_b, lr=1e-4, betas=(0.9, 0.98))
        for _ in range(2):                                   # two optimizer steps: explicit tape vs autograd tape from the same state
            la = float(training.training_step(model, flat, opt, batch, use_tape=True))
            lb = float(training.training_step(model_b, flat_b, opt_b, batch, use_tape=False))
            assert abs(la - lb) <= 1e-5 * abs(lb), (la, lb)
        # Adam turns gradients into +-lr steps: parameters whose gradient is round-off noise may step differently, all others agree
        close = ((flat.flat - flat_b.flat).abs() <= 1e-6).float().mean()
        assert float(close) >= 0.98, float(close)
    finally:
        ag.set_precision("f32")
        training.BF16_STORAGE = keep_storage


@pytest.mark.parametrize("case", [(1, 2, 14, 37, 45, 3), (2, 14, 14, 20, 33, 3), (1, 28, 56, 16, 40, 3), (1, 56, 28, 19, 21, 3), (1, 64, 192, 12, 40, 1),
                                  (1, 192, 96, 9, 33, 1), (1, 288, 144, 10, 12, 3), (1, 18, 36, 640, 372, 3), (1, 7, 5, 8, 32, 3), (3, 33, 130, 5, 7, 1), (1, 4, 16, 20, 18, 5), (1, 16, 16, 20, 18, 3, 2), (2, 6, 24, 13, 35, 5, 2)],
                         ids=lambda c: f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_k{c[5]}" + (f"d{c[6]}" if len(c) > 6 else ""))
def test_conv_wgrad_any_channels(dev, case):
    """The generic weight-gradient kernel (k_conv_wgrad_gen: NormUnet 14 / 28 / 56 ... channels, gate convolutions into 3 F channels; every
    Cout other than 64 and <= 4) against autograd in float64, zero and replicate padding, with and without accumulation."""
    import torch.nn.functional as F
    from mridc_amd import ops
    from tests._util import rel_l2
    B, Cin, Cout, H, W, k = case[:6]
    dil = case[6] if len(case) > 6 else 1
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    p = dil * (k - 1) // 2
    for mode, pm in (("constant", ops.PAD_ZERO), ("replicate", ops.PAD_REPLICATE)):
        w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        xp = F.pad(x.double(), (p, p, p, p), mode=mode) if p else x.double()
        (ref,) = torch.autograd.grad(F.conv2d(xp, w, dilation=dil), w, dy.double())
        got = ops.conv_wgrad(x.to(dev), dy.to(dev), k, dil, pm)
        assert rel_l2(got, ref) <= 2e-6, (mode, rel_l2(got, ref))
        if p == 0:
            break
    acc = torch.ones(Cout, Cin, k, k, device=dev)
    ops.conv_wgrad(x.to(dev), dy.to(dev), k, dil, pm, out=acc, accumulate=True)
    assert rel_l2(acc.cpu().double() - 1.0, ref) <= 5e-6


@pytest.mark.parametrize("case", [("backward", False, "SENSE", True), ("ortho", True, "RSS", False)], ids=lambda c: f"{c[0]}_{c[2]}")
def test_e2evn_training_gradients_vs_oracle_autograd(dev, case):
    """SURVEY 8 row T, E2EVN (vn.py:94-142 under the reference's trainer): loss and EVERY parameter gradient of a 2-cascade VarNet
    (NormUnet 8 channels, 2 pooling levels, odd sizes so the reflect pad / crop branches run) on the HIP training path (mridc_amd/diff.py:
    convolution, transposed-convolution and FFT backward on the kernels) against torch autograd of the CPU oracle."""
    import oracle
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    norm, centered, comb, normalize = case
    cfg = dict(num_cascades=2, channels=8, pooling_layers=2, padding_size=11, normalize=normalize, no_dc=False, use_sens_net=False,
               fft_centered=centered, fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1, coil_combination_method=comb,
               train_loss_fn="l1", val_loss_fn="l1")
    torch.manual_seed(5)
    model = VarNet(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("dc_weight"):
                p_.fill_(0.7)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # (slice 2 with seed 5 sits on a sign boundary of the l1 / |re| + |im| chain in the RSS case: a 1e-7 change of the forward arithmetic moves
    # one gradient by 1e-2 there, 3e-6 everywhere else -- five other seed / slice pairs measured; the RSS case uses slice 3)
    s = synthetic.make_slice(3, 26, 21, slice_idx=2 if comb == "SENSE" else 3)
    y = s["y"] * 50.0                                   # O(1) magnitudes through the normalisations
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    ref_out = oracle.models.varnet_forward(p, cfg, y, s["sensitivity_maps"], s["mask"], None, s["target"])
    ref_mag = torch.abs(ref_out) if comb == "SENSE" else torch.view_as_real(ref_out).abs().sum(-1)
    from tests._util import detie_terms
    tgt = detie_terms(s["target"] * 50.0, ref_mag.detach().unsqueeze(0), 1e-3 * float(ref_mag.detach().abs().max()))    # no l1 term on a sign boundary
    ref_loss = (ref_mag - tgt).abs().mean()
    ref_loss.backward()
    model = model.to(dev).train()
    out = model(y.to(dev), s["sensitivity_maps"].to(dev), s["mask"].to(dev), None, s["target"].to(dev))
    mag = torch.abs(out) if comb == "SENSE" else torch.view_as_real(out).abs().sum(-1)
    loss = (mag - tgt.to(dev)).abs().mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach())), (float(loss.detach()), float(ref_loss.detach()))
    checked = 0
    for name, prm in model.named_parameters():
        ref = p[name].grad
        if name == "dc_weight":                        # the model-level parameter is unused by forward (vn.py:91)
            assert ref is None and prm.grad is None
            continue
        assert ref is not None and prm.grad is not None, name
        assert_close(prm.grad, ref, 2e-3, f"gradient of {name}")
        checked += 1
    assert checked == len(list(model.parameters())) - 1
    # inference is unchanged: eval + no_grad takes the fused kernels and agrees with the recorded forward
    model.eval()
    with torch.no_grad():
        out2 = model(y.to(dev), s["sensitivity_maps"].to(dev), s["mask"].to(dev), None, s["target"].to(dev))
    assert_close(torch.view_as_real(out2), torch.view_as_real(out.detach()), 1e-4, "train vs eval forward")


@pytest.mark.parametrize("cell", ["GRU", "MGU"])
def test_gated_rim_training_gradients_vs_oracle_autograd(dev, cell):
    """RIM with ConvGRU / ConvMGU cells (rim_block.py:217-249, rnn_cells.py:112-127 / 249-261; the reference's base_rim configuration):
    gradients of every parameter of one no_dc cascade on the HIP training path against torch autograd of the CPU oracle."""
    import oracle
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    kw = dict(recurrent_layer=cell, conv_filters=[16, 16, 2], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1], conv_bias=[True, True, False],
              recurrent_filters=[16, 16, 0], recurrent_kernels=[1, 1, 0], recurrent_dilations=[1, 1, 0], recurrent_bias=[True, True, False],
              depth=2, time_steps=4, conv_dim=2, no_dc=True, fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1)
    torch.manual_seed(11)
    blk = RIMBlock(**kw)
    state = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    s = synthetic.make_slice(3, 20, 18, slice_idx=0)
    y, S, m = s["y"] * 50.0, s["sensitivity_maps"], s["mask"]
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    cfg = oracle.rim.RIMConfig(**kw)
    etas_ref, _ = oracle.rim.rim_block_forward(p, cfg, y, y, S, m, None, None, 1.0, False)
    w = torch.linspace(0.2, 1.0, len(etas_ref))
    ref_loss = sum(wi * e.abs().mean() for wi, e in zip(w, etas_ref))
    ref_loss.backward()
    blk = blk.to(dev).train()
    etas, _ = blk(y.to(dev), y.to(dev), S.to(dev), m.to(dev), None, None, 1.0, False)
    loss = sum(wi * e.abs().mean() for wi, e in zip(w.tolist(), etas))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach()))
    checked = 0
    for name, prm in blk.named_parameters():
        ref = p[name].grad
        if name.endswith("dc_weight"):
            continue
        assert ref is not None and prm.grad is not None, name
        assert_close(prm.grad, ref, 2e-3, f"{cell}: gradient of {name}")
        checked += 1
    assert checked >= 9


@pytest.mark.parametrize("shape", [(1, 2, 20, 18), (2, 4, 33, 7), (1, 2, 640, 380)], ids=lambda s: "x".join(map(str, s)))
def test_group_norm_unnorm_and_padding_backward_vs_float64_autograd(dev, shape):
    """NormUnet's frame (unet_block.py:71-111): group norm with the unbiased std, zero padding, [a stand-in for the U-Net], crop, un-normalisation with the SAME
    statistics -- the gradient reaches x through the normalised tensor, through the mean and through the std.  mrx_group_norm_bwd / mrx_pad2d against torch autograd of the
    reference's formulas in float64."""
    from mridc_amd import diff
    B, C, H, W = shape
    groups = 2
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, C, H, W, generator=g) * 3.0 + 0.7
    wgt = torch.randn(B, C, H + 5, W + 3, generator=g)                 # the "network": an elementwise non-linearity with position-dependent weights
    dy = torch.randn(B, C, H, W, generator=g)

    def frame(x_, w_, norm, unnorm, pad):
        xn, mean, std = norm(x_, groups)
        p = pad(xn, 2, 3, 1, 2, 0)
        q = p * w_ + 0.25 * p * p
        c = pad(q, -2, -3, -1, -2, 0)
        return unnorm(c, mean, std, groups)

    def norm64(x_, G):
        b, c, h, w = x_.shape
        xg = x_.reshape(b, G, -1)
        mean, std = xg.mean(-1, keepdim=True), xg.std(-1, keepdim=True)
        return ((xg - mean) / std).reshape(b, c, h, w), mean, std

    def unnorm64(x_, mean, std, G):
        b, c, h, w = x_.shape
        return (x_.reshape(b, G, -1) * std + mean).reshape(b, c, h, w)

    def pad64(x_, t, b_, l, r, mode):
        return torch.nn.functional.pad(x_, (l, r, t, b_))

    xr = x.double().requires_grad_(True)
    ref = frame(xr, wgt.double(), norm64, unnorm64, pad64)
    ref.backward(dy.double())
    xd = x.to(dev).requires_grad_(True)
    got = frame(xd, wgt.to(dev), diff.group_norm, diff.group_unnorm, diff.pad2d)
    assert_close(got, ref.detach().float(), 2e-6, "frame forward")
    got.backward(dy.to(dev))
    assert_close(xd.grad, xr.grad.float(), 5e-6, "gradient through the normalised tensor, the mean and the std")
    # the normalisation alone, statistics unused downstream (their gradients are None)
    xd2 = x.to(dev).requires_grad_(True)
    diff.group_norm(xd2, groups)[0].backward(dy.to(dev))
    xr2 = x.double().requires_grad_(True)
    norm64(xr2, groups)[0].backward(dy.double())
    assert_close(xd2.grad, xr2.grad.float(), 5e-6, "group norm alone")


@pytest.mark.parametrize("cell", ["GRU", "MGU"])
@pytest.mark.parametrize("shape", [(1, 16, 20, 18), (2, 5, 7, 33), (1, 64, 64, 48)], ids=lambda s: "x".join(map(str, s)))
def test_gate_backward_kernels_vs_float64_autograd(dev, cell, shape):
    """mrx_gru_gates_bwd / mrx_mgu_gates_bwd: the three gradients of the gate math (rnn_cells.py:118-127, :255-261) in one launch each, against torch
    autograd of the same formulas in float64; the forward of diff.gru_gates / mgu_gates is the inference kernel's."""
    from mridc_amd import diff, ops
    B, F, H, W = shape
    G = 3 if cell == "GRU" else 2
    g = torch.Generator().manual_seed(F + H)
    ih, hh = torch.randn(B, G * F, H, W, generator=g) * 1.5, torch.randn(B, G * F, H, W, generator=g) * 1.5
    hx, dy = torch.randn(B, F, H, W, generator=g), torch.randn(B, F, H, W, generator=g)
    a, b, c = (t.double().requires_grad_(True) for t in (ih, hh, hx))
    if cell == "GRU":
        (i_r, i_z, i_n), (h_r, h_z, h_n) = a.chunk(3, 1), b.chunk(3, 1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        out = n * (1 - z) + z * c
    else:
        (i_f, i_c), (h_f, h_c) = a.chunk(2, 1), b.chunk(2, 1)
        f = torch.sigmoid(i_f + h_f)
        cc = torch.tanh(i_c + f * h_c)
        out = cc + f * (c - cc)
    out.backward(dy.double())
    x, y_, h_ = (t.to(dev).requires_grad_(True) for t in (ih, hh, hx))
    got = (diff.gru_gates if cell == "GRU" else diff.mgu_gates)(x, y_, h_)
    assert got.grad_fn is not None
    assert_close(got, out.detach().float(), 2e-6, f"{cell} gates forward")
    assert torch.equal(got.detach(), (ops.gru_gates if cell == "GRU" else ops.mgu_gates)(x.detach(), y_.detach(), h_.detach()))
    got.backward(dy.to(dev))
    for name, mine, ref in (("d ih", x.grad, a.grad), ("d hh", y_.grad, b.grad), ("d h", h_.grad, c.grad)):
        assert_close(mine, ref.float(), 3e-6, f"{cell} {name}")
    with pytest.raises(ValueError):
        ops.gru_gates_bwd(dy.to(dev), ih.to(dev)[:, :F], hh.to(dev), hx.to(dev))


@pytest.mark.parametrize("seed", list(range(8)))
def test_absl1_loss_gradient_includes_the_path_through_the_maximum(dev, seed):
    """mean |target - |p| / max|p|| (cirim.py:218-237): the gradient has one large term at the arg-max pixel (the path through the max).
    mrx_absl1_loss_bwd finds that pixel by comparing its own modulus with mrx_max_abs's, so both must round identically (they are formed by
    the same fmul / fmul / fadd / correctly-rounded-sqrt sequence whatever each file's contraction mode).  Against torch autograd in
    float64 on several seeds; the arg-max entry is checked on its own, because the whole-vector norm would hide a dropped term."""
    from mridc_amd import autograd as ag
    g = torch.Generator().manual_seed(100 + seed)
    p = (torch.randn(1, 37, 41, 2, generator=g) * 3.0 + 0.5)
    tgt = torch.rand(1, 37, 41, generator=g)
    pr = p.double().requires_grad_(True)
    mag = (pr ** 2).sum(-1).sqrt()
    ref = (tgt.double() - mag / mag.max()).abs().mean()
    ref.backward()
    pd = p.to(dev).requires_grad_(True)
    loss = ag.AbsL1Loss.apply(pd, tgt.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
    got, want = pd.grad.cpu().double(), pr.grad
    assert rel_l2(got, want) <= 1e-5
    idx = int(mag.detach().reshape(-1).argmax())
    gw, gg = want.reshape(-1, 2)[idx], got.reshape(-1, 2)[idx]
    assert float(gw.norm()) > 10 * float(want.reshape(-1, 2).norm(dim=1).median())       # the max term really dominates that pixel
    assert float((gg - gw).norm()) <= 1e-4 * float(gw.norm()), (gg, gw)


def _tape_rank(rank, world, port, tmp):
    """One rank of the two-process explicit-tape step (a fresh process: it initialises the GPU itself)."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
    torch.manual_seed(3)
    model = CIRIM(cfg).to(dev)
    s = synthetic.make_slice(3, 24, 20, slice_idx=1 + rank)            # every rank its own slice
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    flat = training.FlatParameters(model)
    assert len(flat.cascade_slices) == 2
    opt = training.AdamFlat(flat, lr=1e-3, betas=(0.9, 0.98))
    loss = training.training_step(model, flat, opt, batch, use_tape=True)   # per-cascade async all-reduce of the flat gradient's slices
    torch.cuda.synchronize()
    torch.save(dict(loss=float(loss), grad=flat.grad.detach().cpu(), params=flat.flat.detach().cpu()), os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_on_the_explicit_tape(dev, tmp_path):
    """SURVEY 8 row T / 8e (base_cirim_train.yaml:175 `strategy: ddp`): `training.training_step(use_tape=True)` with TWO ranks (two fresh
    processes sharing the one GPU, gloo process group on device tensors): each cascade's slice of the flat gradient is all-reduced
    asynchronously as soon as its backward ends.  The reduced gradient equals the sum of the two slices' single-rank gradients, and both
    ranks hold identical parameters after the Adam step."""
    import socket
    import torch.multiprocessing as mp
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tape_rank, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(600)
        assert p_.exitcode == 0, p_.exitcode
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    assert torch.equal(got[0]["grad"], got[1]["grad"]) and torch.equal(got[0]["params"], got[1]["params"])
    # single-rank reference in this process: the gradients of the two slices, summed
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
    total = None
    for r in range(2):
        torch.manual_seed(3)
        model = CIRIM(cfg).to(dev).train()
        s = synthetic.make_slice(3, 24, 20, slice_idx=1 + r)
        batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
        flat = training.FlatParameters(model)
        flat.zero_grad()
        training.cirim_forward_backward(model, batch, "f32")
        total = flat.grad.detach().cpu().clone() if total is None else total + flat.grad.detach().cpu()
    assert_close(got[0]["grad"], total, 1e-6, "two-rank reduced gradient vs the sum of the single-rank gradients")


@pytest.mark.parametrize("shape", [(2, 3, 9, 7), (1, 5, 130, 90), (1, 2, 640, 384)], ids=lambda s: "x".join(map(str, s)))
def test_unet_backward_steps_on_hip_kernels(dev, shape):
    """csrc/diff_bwd.hip (round 4: these ran as torch device ops inside the backward): activation, InstanceNorm2d + LeakyReLU, avg_pool2d adjoint and
    the pixel-unshuffle of ConvTranspose2d's gradients against torch autograd in float64 (unet_block.py:251-258, 189-230)."""
    from mridc_amd import diff, ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, C, H, W, generator=g) * 2.0 + 0.3
    dy = torch.randn(B, C, H, W, generator=g)
    # activation
    y = F.leaky_relu(x, 0.2)
    assert torch.equal(ops.act_bwd(dy.to(dev), y.to(dev), ops.ACT_LEAKY, 0.2).cpu(), torch.where(y > 0, dy, dy * 0.2))
    assert torch.equal(ops.act_bwd(dy.to(dev), F.relu(x).to(dev), ops.ACT_RELU).cpu(), torch.where(x > 0, dy, torch.zeros(())))
    # instance norm + LeakyReLU / no activation
    for act, slope in ((ops.ACT_LEAKY, 0.2), (ops.ACT_NONE, 0.0)):
        xd = x.double().requires_grad_(True)
        n = F.instance_norm(xd, eps=1e-5)
        (F.leaky_relu(n, slope) if act == ops.ACT_LEAKY else n).backward(dy.double())
        xg = x.to(dev).requires_grad_(True)
        out = diff.instance_norm_act(xg, 1e-5, act, slope)
        out.backward(dy.to(dev))
        assert_close(xg.grad, xd.grad.float(), 2e-5, f"instance norm backward, act {act}")
    # average pooling (odd sizes leave the last row / column unpooled)
    xd = x.double().requires_grad_(True)
    po = F.avg_pool2d(xd, 2)
    gp = torch.randn(po.shape, generator=g)
    po.backward(gp.double())
    assert_close(ops.avg_pool2x2_bwd(gp.to(dev), H, W), xd.grad.float(), 1e-6, "avg_pool2d adjoint")
    # pixel unshuffle
    if H % 2 == 0 and W % 2 == 0:
        assert torch.equal(ops.pixel_unshuffle2(x.to(dev)).cpu(), F.pixel_unshuffle(x, 2))


@pytest.mark.parametrize("case", [("backward", False, 21), ("ortho", True, 372), ("forward", True, 30)], ids=lambda c: f"{c[0]}_W{c[2]}")
def test_coil_operator_and_dc_backward_on_hip_kernels(dev, case):
    """diff.sens_expand / sens_reduce / dc_combine (vn_block.py:51-119): forward on the fused kernels, backward = the opposite transform + ONE pointwise
    pass (csrc/diff_bwd.hip: mrx_sens_expand_bwd_pw, mrx_cmul_bcast, mrx_dc_combine_bwd), against float64 autograd of the written-out reference forms
    -- gradients w.r.t. the image / k-space, the sensitivity maps (the E2EVN sensitivity network is trained through them) and dc_weight."""
    import oracle
    from mridc_amd import diff
    norm, centered, W = case
    B, C, H = 2, 3, 10
    g = torch.Generator().manual_seed(W)
    x, S = torch.randn(B, H, W, 2, generator=g), torch.randn(B, C, H, W, 2, generator=g)
    k, ref = torch.randn(B, C, H, W, 2, generator=g), torch.randn(B, C, H, W, 2, generator=g)
    mask = (torch.rand(1, 1, H, W, 1, generator=g) < 0.4).float()
    w = torch.tensor([0.7])
    g1, g2, g3 = torch.randn(B, C, H, W, 2, generator=g), torch.randn(B, H, W, 2, generator=g), torch.randn(B, C, H, W, 2, generator=g)
    cm, cj = oracle.utils.complex_mul, oracle.utils.complex_conj

    def leaves(dtype, device):
        return [t.to(device=device, dtype=dtype).requires_grad_(True) for t in (x, S, k, w)]

    # float64 reference (the reference's own composition of torch ops)
    xr, Sr, kr, wr = leaves(torch.float64, "cpu")
    e_r = oracle.fft.fft2(cm(xr.unsqueeze(1), Sr), centered, norm, [-2, -1])
    r_r = cm(oracle.fft.ifft2(kr, centered, norm, [-2, -1]), cj(Sr)).sum(1)
    zero = torch.zeros(1, 1, 1, 1, 1, dtype=torch.float64)
    d_r = kr - torch.where(mask.bool(), kr - ref.double(), zero) * wr - e_r
    ((e_r * g1.double()).sum() + (r_r * g2.double()).sum() + (d_r * g3.double()).sum()).backward()
    # HIP path
    xd, Sd, kd, wd = leaves(torch.float32, dev)
    e_d = diff.sens_expand(xd, Sd, centered, norm, [-2, -1])
    r_d = diff.sens_reduce(kd, Sd, centered, norm, [-2, -1])
    d_d = diff.dc_combine(kd, kd, ref.to(dev), mask.to(dev), wd, e_d)
    assert_close(e_d, e_r.float(), 1e-5, "sens_expand forward")
    assert_close(r_d, r_r.float(), 1e-5, "sens_reduce forward")
    assert_close(d_d, d_r.float(), 1e-5, "dc_combine forward")
    ((e_d * g1.to(dev)).sum() + (r_d * g2.to(dev)).sum() + (d_d * g3.to(dev)).sum()).backward()
    for name, got, want in (("image", xd, xr), ("maps", Sd, Sr), ("k-space", kd, kr), ("dc_weight", wd, wr)):
        assert_close(got.grad, want.grad.float(), 2e-5, f"gradient w.r.t. {name}")
    # base and pred distinct tensors, nothing else requiring a gradient
    b2 = torch.randn(B, C, H, W, 2, generator=g)
    bd, br = b2.to(dev).requires_grad_(True), b2.double().requires_grad_(True)
    diff.dc_combine(bd, k.to(dev), ref.to(dev), mask.to(dev), w.to(dev), ref.to(dev)).backward(g3.to(dev))
    (br - torch.where(mask.bool(), k.double() - ref.double(), zero) * w.double() - ref.double()).backward(g3.double())
    assert_close(bd.grad, br.grad.float(), 1e-6, "gradient w.r.t. a separate base")


def test_graphed_model_training_step_follows_the_eager_one(dev):
    """training.GraphedModelStep (forward + loss + backward of an E2EVN as one hipGraph replay, weight packs inside the graph): three steps on three
    different slices leave the same parameters as three eager model_training_step calls -- the replay sees refilled batches AND the weights the
    optimizer has updated in between."""
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    cfg = dict(num_cascades=2, channels=8, pooling_layers=2, padding_size=11, normalize=True, no_dc=False, use_sens_net=False, fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1,
               coil_combination_method="SENSE", train_loss_fn="l1", val_loss_fn="l1")
    C, H, W = 4, 40, 372                                  # (W = 372: the prime-factor FFT route with its per-slice prepared maps)
    batches = []
    for i in range(3):
        s = synthetic.make_slice(C, H, W, slice_idx=i)
        batches.append({k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")})
    finals = []
    for graphed in (False, True):
        torch.manual_seed(3)
        model = VarNet(cfg).to(dev)
        flat = training.FlatParameters(model)
        opt = training.AdamFlat(flat, lr=1e-3)
        step = training.GraphedModelStep(model, flat, opt, batches[0]) if graphed else None
        if graphed:                                       # the constructor's warm-up steps ran forward / backward only: no optimizer step, same weights
            assert opt.steps == 0
        losses = []
        for b in batches:
            losses.append(float(step(b) if graphed else training.model_training_step(model, flat, opt, b)))
        finals.append((flat.flat.clone(), losses))
    (w_e, l_e), (w_g, l_g) = finals
    assert l_e == pytest.approx(l_g, rel=1e-5), (l_e, l_g)
    assert_close(w_g, w_e, 1e-5, "parameters after three graphed steps vs three eager steps")
    assert float((w_e - finals[0][0]).abs().max()) == 0.0


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graphed_cirim_training_step_follows_the_eager_one(dev, precision):
    """training.GraphedCirimStep (the explicit tape as one hipGraph replay; in bf16 the side stream's weight gradients are parallel branches of the
    graph): three steps on three slices leave the parameters of three eager training_step calls."""
    from mridc_amd import autograd as ag
    from mridc_amd import synthetic, training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2, recurrent_layer="IndRNN")
    batches = []
    for i in range(3):
        s = synthetic.make_slice(4, 24, 372, slice_idx=i)
        batches.append({k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")})
    keep = ag.PRECISION
    ag.set_precision(precision)
    try:
        finals = []
        for graphed in (False, True):
            torch.manual_seed(4)
            model = CIRIM(cfg).to(dev)
            flat = training.FlatParameters(model)
            opt = training.AdamFlat(flat, lr=1e-3)
            step = training.GraphedCirimStep(model, flat, opt, batches[0]) if graphed else None
            losses = [float(step(b) if graphed else training.training_step(model, flat, opt, b)) for b in batches]
            finals.append((flat.flat.clone(), losses))
    finally:
        ag.set_precision(keep)
    (w_e, l_e), (w_g, l_g) = finals
    assert l_e == pytest.approx(l_g, rel=1e-5), (l_e, l_g)
    assert_close(w_g, w_e, 1e-5, "parameters after three graphed steps vs three eager steps")
