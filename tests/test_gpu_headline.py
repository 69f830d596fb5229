"""GPU parity AT THE HEADLINE SHAPES: the exact kernels bench.py's timed loop runs (BASELINE.json configs[1], [2], [4]) against the
oracle on the same seeded inputs.

The small golden fixtures (W <= 48) take the runtime FFT plans, one coil chunk and single-tile convolutions; the timed loop at
15 x 640 x 372 takes the compile-time 372-point plan, four coil chunks whose partial sums are finished by the first RIM layer's tile
loader (`mrx_llg_hinv_parts` + `mrx_rim_layer_indrnn_packed_llg`, rim_block.py:217-249 / rim_utils.py:11-67), the 960-tile
persistent Winograd loop with the XCD band order and the 4-pixel final conv.  These tests pin exactly that path.
Tolerances (fp32): operators 1e-5, one 8-step block 2e-5, the 64-step chain 1e-4 and SSIM >= 0.9999."""
import os

import pytest
import torch

import oracle
from mridc_amd import synthetic
from tests._util import T, assert_close, meta, rel_l2, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    return torch.device("cuda:0")


def _problem(B, C, H, W, seed, centered, norm, mask_dtype=torch.bool):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 1, H, W, 2, generator=g)
    S = torch.randn(B, C, H, W, 2, generator=g)
    S = S / oracle.utils.complex_abs_sq(S).sum(1, keepdim=True).sqrt().unsqueeze(-1)
    k = oracle.fft.fft2(oracle.utils.complex_mul(img, S), centered, norm)
    k = k / k.abs().max()
    m = torch.from_numpy(synthetic.random_mask_1d(W, seed=seed)).reshape(1, 1, 1, W, 1)
    y = k * m
    eta = torch.randn(B, H, W, 2, generator=g) * 0.3
    return y, S, (m if mask_dtype == torch.bool else m.to(mask_dtype)), eta


def _layer1(seed, F=64):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(F, 4, 5, 5, generator=g) * 0.15
    b = torch.randn(F, generator=g) * 0.1
    wi = torch.randn(F, F, 1, 1, generator=g) * 0.2
    bi = torch.randn(F, generator=g) * 0.1
    hh = torch.randn(1, F, 1, 1, generator=g) * 0.5
    return w, b, wi, bi, hh


# (B, C, H, W, centered, norm, sigma, mask dtype, expect the deferred route)
DEFERRED_CASES = [
    (1, 15, 640, 372, False, "backward", 1.0, torch.bool, True),       # the headline launch itself
    (1, 15, 64, 372, True, "ortho", 0.7, torch.uint8, True),
    (2, 5, 40, 372, False, "backward", 1.3, torch.float32, True),      # B > 1, two chunks (4 + 1 coils), ragged tile rows
    (1, 32, 44, 372, True, "forward", 1.0, torch.bool, True),          # eight chunks, rows not a multiple of the 8-row tile
    (1, 15, 72, 320, False, "backward", 0.9, torch.bool, True),
    (3, 15, 24, 320, True, "ortho", 1.0, torch.uint8, True),       # B > 1; 320: six coils per workgroup -> 3 chunks
    (2, 5, 24, 320, False, "ortho", 1.0, torch.bool, False),         # C <= 6: one chunk, nothing to defer
    (1, 32, 256, 256, False, "backward", 1.1, torch.bool, True),       # the qCIRIM slice shape
    (1, 15, 48, 256, True, "ortho", 1.0, torch.float32, True),
    (2, 15, 512, 372, False, "backward", 1.0, torch.bool, False),      # H * B = 1024: the coil sum stays inside one workgroup
    (1, 15, 1024, 256, False, "none", 1.0, torch.bool, False),
]


@pytest.mark.parametrize("case", DEFERRED_CASES, ids=lambda c: f"B{c[0]}C{c[1]}_{c[2]}x{c[3]}_{'c' if c[4] else 'n'}_{c[5]}")
def test_llg_parts_and_layer1_loader(dev, case):
    """(a) `mrx_llg_hinv_parts` + `mrx_rim_layer_indrnn_packed_llg` vs oracle.rim.log_likelihood_gradient + conv5x5 + IndRNN."""
    from mridc_amd import ops
    B, C, H, W, centered, norm, sigma, mdt, want_defer = case
    y, S, mask, eta = _problem(B, C, H, W, 1000 + H + W + C, centered, norm, mdt)
    w, b, wi, bi, hh = _layer1(77)
    h_prev = torch.randn(B, 64, H, W, generator=torch.Generator().manual_seed(5)).relu()
    with torch.no_grad():
        g_ref = oracle.rim.log_likelihood_gradient(eta, y, S, mask, sigma, centered, norm, [-2, -1], 1).contiguous()
        a_ref = oracle.rim.conv_nonlinear(g_ref, w, b, 5, 1, "relu")
        h_ref = oracle.rim.indrnn_cell(a_ref, h_prev, wi, bi, hh, 1, 1)
    yd, Sd, md, ed, hp = y.to(dev), S.to(dev), mask.to(dev), eta.to(dev), h_prev.to(dev)
    wd, bd, wid, bid, hhd = (t.to(dev) for t in (w, b, wi, bi, hh))
    yt = ops.llg_prepare(yd, centered, norm)
    packed = ops.rim_layer_pack(wd, wid)
    out4, part, nparts = ops.llg_hinv_parts(ed, yt, Sd, md, sigma, centered, norm)
    assert (nparts > 0) == want_defer, f"nparts = {nparts}"
    full = ops.llg_hinv(ed, yt, Sd, md, sigma, centered, norm)                     # the same gradient with the combine done
    assert_close(full, g_ref, 1e-5, "llg_hinv vs oracle")
    h_plain = ops.rim_layer_indrnn_packed(full, packed, 64, 5, 1, bd, bid, hhd, hp)
    assert_close(h_plain, h_ref, 1e-5, "layer 1 (complete gradient) vs oracle")
    if nparts > 0:
        assert nparts == -(-C // 4) or W != 372                                     # 372: four coils per workgroup
        h_def = ops.rim_layer_indrnn_packed_llg(ed, part, nparts, sigma, packed, 64, 5, 1, bd, bid, hhd, hp)
        assert_close(h_def, h_ref, 1e-5, "layer 1 reading the coil-chunk partial sums vs oracle")
        assert rel_l2(h_def, h_plain) <= 2e-6
        h0 = ops.rim_layer_indrnn_packed_llg(ed, part, nparts, sigma, packed, 64, 5, 1, bd, bid, hhd, None)   # zero initial state
        with torch.no_grad():
            h0_ref = oracle.rim.indrnn_cell(a_ref, torch.zeros_like(h_prev), wi, bi, hh, 1, 1)
        assert_close(h0, h0_ref, 1e-5, "layer 1 (deferred, zero state) vs oracle")
    else:
        assert_close(out4, g_ref, 1e-5, "llg_hinv_parts complete output vs oracle")
    if W == 372:
        # the prime-factor kernel (mrx_llg372: lane-ordered operands prepared once per slice), complete and deferred forms
        assert ops.llg372_supported(yt, md)
        keep = ops.LLG372_NO_Y
        try:
            for no_y in (True, False):       # the data as one constant plane per slice (default) / read by every step
                ops.LLG372_NO_Y = no_y
                op = ops.llg372_prepare(yt, Sd, md, centered, norm if no_y else None)
                assert_close(ops.llg372(ed, op, sigma, norm), g_ref, 1e-5, f"llg372 vs oracle (no_y={no_y})")
                part3, n3 = ops.llg372(ed, op, sigma, norm, parts=True)
                assert n3 == -(-C // 5) + int(no_y)
                h3 = ops.rim_layer_indrnn_packed_llg(ed, part3, n3, sigma, packed, 64, 5, 1, bd, bid, hhd, hp)
                assert_close(h3, h_ref, 1e-5, f"layer 1 reading the llg372 partial sums vs oracle (no_y={no_y})")
            # prepared without a normalization: the constant plane is made by the first call, and again when the normalization changes
            ops.LLG372_NO_Y = True
            op = ops.llg372_prepare(yt, Sd, md, centered)
            assert op.const_norm is None
            assert_close(ops.llg372(ed, op, sigma, norm), g_ref, 1e-5, "llg372 vs oracle (lazy constant plane)")
            assert op.const_norm is not None
            # the linear part alone (training's adjoint): g(eta) - g(0), and self-adjoint
            lin = op.linear_part()
            g1, g0 = ops.llg372(ed, op, sigma, norm)[:, 2:], ops.llg372(torch.zeros_like(ed), op, sigma, norm)[:, 2:]
            a1 = ops.llg372(ed, lin, sigma, norm)[:, 2:]
            assert_close(a1, (g1 - g0).cpu(), 1e-4, "linear part = g(eta) - g(0)")      # (a difference of two fp32 results)
            e2 = torch.randn_like(ed)
            a2 = ops.llg372(e2, lin, sigma, norm)[:, 2:]
            d12 = float((a1.double() * e2.permute(0, 3, 1, 2).double()).sum())
            d21 = float((a2.double() * ed.permute(0, 3, 1, 2).double()).sum())
            assert abs(d12 - d21) <= 1e-4 * max(abs(d12), abs(d21), 1e-30), (d12, d21)
            # the final convolution's tap gather folded into the gradient launch (mrx_llg372_gather): bit-identical to the two launches
            taps = torch.randn(B, 18, H, W, generator=torch.Generator().manual_seed(5)).to(dev) * 0.1
            bfin = torch.tensor([0.03, -0.02], device=dev)
            for bias in (bfin, None):
                eta2 = ops.rim_final_gather(taps, bias, ed)
                want_parts, n_w = ops.llg372(eta2, op, sigma, norm, parts=True)
                want_parts = want_parts.clone()
                got_parts, n_g, eta_g = ops.llg372_gather(ed, taps, bias, op, sigma, norm)
                assert n_g == n_w and torch.equal(eta_g, eta2) and torch.equal(got_parts, want_parts)
        finally:
            ops.LLG372_NO_Y = keep


def test_llg372_batched_mask_and_ragged_coils(dev):
    """mrx_llg372 with one mask per batch element ([B,1,1,W,1]) and coil counts that leave the last task partly empty."""
    from mridc_amd import ops
    for B, C, H, centered, norm in ((2, 7, 9, True, "ortho"), (3, 1, 5, False, "backward"), (1, 16, 33, False, "forward")):
        y, S, mask, eta = _problem(B, C, H, 372, 4000 + C, centered, norm)
        mB = torch.cat([torch.roll(mask, 7 * i, dims=3) for i in range(B)], 0)
        k = oracle.fft.fft2(oracle.utils.complex_mul(torch.randn(B, 1, H, 372, 2, generator=torch.Generator().manual_seed(C)), S),
                            centered, norm)
        y = (k / k.abs().max()) * mB
        with torch.no_grad():
            g_ref = oracle.rim.log_likelihood_gradient(eta, y, S, mB, 0.8, centered, norm, [-2, -1], 1).contiguous()
        yd, Sd, md, ed = y.to(dev), S.to(dev), mB.to(dev), eta.to(dev)
        yt = ops.llg_prepare(yd, centered, norm)
        op = ops.llg372_prepare(yt, Sd, md, centered)
        assert_close(ops.llg372(ed, op, 0.8, norm), g_ref, 1e-5, f"llg372 B={B} C={C}")
        assert_close(ops.llg_hinv(ed, yt, Sd, md, 0.8, centered, norm), g_ref, 1e-5, f"llg_hinv B={B} C={C}")


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


@pytest.mark.parametrize("scale", [1.0, 5.0], ids=["reference_init", "x5_recurrent_weights"])
def test_full_size_rim_block_winograd_on_and_off(dev, scale):
    """(b) one whole RIMBlock (8 steps, IndRNN 64) at 1 x 15 x 640 x 372 vs the oracle, with the Winograd layer-2 kernel and with the
    direct one; both hidden states too."""
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    cfg, model, sd = _cirim(dict(num_cascades=1), scale)
    d = synthetic.make_slice(15, 640, 372, slice_idx=3)
    rc = oracle.rim.RIMConfig(**{k: cfg[k] for k in ("recurrent_layer", "conv_filters", "conv_kernels", "conv_dilations", "conv_bias",
                                                     "recurrent_filters", "recurrent_kernels", "recurrent_dilations", "recurrent_bias",
                                                     "depth", "no_dc", "fft_centered", "fft_normalization", "spatial_dims", "coil_dim")},
                              time_steps=8)
    p = {k[len("cirim.0."):]: v for k, v in sd.items() if k.startswith("cirim.0.")}
    with torch.no_grad():
        ref, ref_hx = oracle.rim.rim_block_forward(p, rc, d["y"], d["y"], d["sensitivity_maps"], d["mask"], None, None, 1.0, False)
    blk = model.cirim[0].to(dev)
    y, S, m = d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev)
    outs = {}
    for wino in (True, False):
        RIMBlock.winograd, keep = wino, RIMBlock.winograd
        try:
            blk._pack_cache.clear()
            with torch.no_grad():
                etas, hx = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
        finally:
            RIMBlock.winograd = keep
        assert len(etas) == 8
        assert_close(torch.stack(etas), torch.stack(ref), 2e-5, f"8-step block, winograd={wino}")
        for j in range(2):
            assert_close(hx[j], ref_hx[j], 2e-5, f"hidden state {j}, winograd={wino}")
        outs[wino] = etas[-1]
    assert rel_l2(outs[True], outs[False]) <= 5e-6


@pytest.mark.parametrize("shape", [(15, 640, 372), (15, 40, 372), (6, 37, 75)])
def test_rim_block_fused_final_and_in_place_state(dev, shape):
    """The two default shortcuts of the IndRNN cascade against their plain forms, whole RIMBlocks (two of them chained, so that the second
    receives the first one's states as the CALLER's): (1) the final convolution's channel contraction in layer 2's tail + tap gather
    (mrx_rim_layer2_sb_taps / mrx_rim_final_gather) vs the stand-alone final kernel: fp32 round-off; (2) hidden states overwritten in place
    from the second step on vs fresh tensors every step: bit-identical -- and a state handed in by the caller must come back untouched."""
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    C, H, W = shape
    cfg, model, sd = _cirim(dict(num_cascades=2), 3.0, seed=2)
    d = synthetic.make_slice(C, H, W, slice_idx=5)
    y, S, m = d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev)
    blk0, blk1 = model.cirim[0].to(dev), model.cirim[1].to(dev)
    res = {}
    keep = (RIMBlock.fused_final, RIMBlock.inplace_state)
    try:
        for fused, inplace in ((True, True), (True, False), (False, True), (False, False)):
            RIMBlock.fused_final, RIMBlock.inplace_state = fused, inplace
            with torch.no_grad():
                etas0, hx0 = blk0(y, y, S, m, None, None, 1.0, keep_eta=False)
                saved = [h.clone() for h in hx0]
                etas1, hx1 = blk1(etas0, y, S, m, etas0[-1], hx0, 1.0, keep_eta=False)
            for a, b in zip(saved, hx0):
                assert torch.equal(a, b), "the caller's hidden state was modified"
            res[(fused, inplace)] = (torch.stack(etas0 + etas1), torch.stack(list(hx1)))
    finally:
        RIMBlock.fused_final, RIMBlock.inplace_state = keep
    for fused in (True, False):
        for k in range(2):
            assert torch.equal(res[(fused, True)][k], res[(fused, False)][k]), f"in-place state changed the result (fused_final={fused})"
    assert rel_l2(res[(True, True)][0], res[(False, True)][0]) <= 5e-6
    assert rel_l2(res[(True, True)][1], res[(False, True)][1]) <= 5e-6


def test_eight_cascades_w372_final_image_ssim(dev):
    """(c) all 8 cascades x 8 steps at 1 x 15 x 64 x 372 (the 372-point plan, four coil chunks, deferred route) vs the oracle; SSIM vs
    ref on the FINAL image through the product's harness (mridc_amd.runner) with the oracle's SSIM as the checker."""
    from mridc_amd import runner
    cfg, model, sd = _cirim({}, 4.0, seed=1)
    d = synthetic.make_slice(15, 64, 372, slice_idx=1)
    with torch.no_grad():
        ref = oracle.models.cirim_forward(sd, cfg, d["y"], d["sensitivity_maps"], d["mask"], None, d["target"])
    model = model.to(dev)
    with torch.no_grad():
        out = next(model(d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev), None, d["target"].to(dev)))
    got = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
    want = torch.view_as_real(torch.stack([torch.stack(c) for c in ref]))
    assert_close(got, want, 1e-4, "8 x 8 chain at W = 372")
    o_gpu, o_ref = runner.postprocess(out[-1][-1], ref[-1][-1].to(dev))
    ssim = runner.metrics_to_dict(runner.slice_metrics(o_gpu, o_ref))["SSIM"]
    c1, c2 = oracle.metrics.postprocess(out[-1][-1].cpu(), ref[-1][-1])
    chk = oracle.metrics.ssim(c2.numpy(), c1.numpy(), maxval=float(c1.max() - c1.min()))
    assert ssim >= 0.9999 and abs(ssim - chk) <= 1e-5, (ssim, chk)


def test_headline_launch_shape_eight_slices_per_launch(dev):
    """bench.py's default launch: EIGHT distinct slices per call (B = 8) at 15 x 640 x 372 -- 3 840 tiles = 15 exact rounds of the persistent layer kernels, the
    batch index inside every kernel's tile decomposition, the batched operand preparation of mrx_llg372 -- one whole cascade (8 time-steps, two weight sets)
    against the oracle slice by slice, and each slice of the batch against the same slice run alone (bit-identical: no kernel mixes batch elements)."""
    for scale in (1.0, 5.0):
        cfg, model, sd = _cirim(dict(num_cascades=1), scale)
        slices = [synthetic.make_slice(15, 640, 372, slice_idx=10 + i) for i in range(8)]
        y = torch.cat([s_["y"] for s_ in slices], 0)
        S = torch.cat([s_["sensitivity_maps"] for s_ in slices], 0)
        tgt = torch.cat([s_["target"] for s_ in slices], 0)
        mask = slices[0]["mask"]
        model = model.to(dev)
        with torch.no_grad():
            out = next(model(y.to(dev), S.to(dev), mask.to(dev), None, tgt.to(dev)))
            one = next(model(y[5:6].to(dev), S[5:6].to(dev), mask.to(dev), None, tgt[5:6].to(dev)))
        assert len(out) == 1 and len(out[0]) == 8 and tuple(out[0][-1].shape) == (8, 640, 372)
        assert torch.equal(out[0][-1][5:6], one[0][-1])
        for i in (0, 3, 7):                                        # (three of the eight on the CPU oracle: ~4 s each)
            with torch.no_grad():
                ref = oracle.models.cirim_forward(sd, cfg, y[i:i + 1], S[i:i + 1], mask, None, tgt[i:i + 1])
            got = torch.view_as_real(torch.stack([o_[i:i + 1] for o_ in out[0]]))
            want = torch.view_as_real(torch.stack(ref[0]))
            assert_close(got, want, 2e-5, f"B = 8 launch, slice {i}, recurrent weights x{scale}")


def test_g19_spec_fixture_and_harness_metrics(golden, dev):
    """The G6 fixture as SURVEY 8c specifies it ([1,15,64,48], 8 cascades, 64 filters; generated by the reference) through the model
    AND through the harness runner: post-processed image + MSE / NMSE / SSIM / PSNR (models/base.py:415-436)."""
    from mridc_amd import runner
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    z = golden("g19_cirim_spec.npz")
    cfg = meta(z, "cfg")
    model = CIRIM(cfg)
    missing, unexpected = model.load_state_dict(weights(z, "w/"), strict=False)
    assert unexpected == [] and missing == ["dc_weight"]
    model = model.to(dev).eval()
    y, S, mask, target = (T(z[k]).to(dev) for k in ("y", "S", "mask", "target"))
    with torch.no_grad():
        out = next(model(y, S, mask, None, target))
    got = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
    assert_close(got, T(z["out"]), 1e-4, "g19 chain")
    r = runner.ReconstructionRunner(model)
    name, sl, pred = r.test_step((y, y, S, mask, None, target, ["vol_a"], 7, 4.0))
    assert (name, sl) == ("vol_a", 7) and pred.shape == (1, 64, 48)
    r.test_step((y, y, S, mask, None, target, ["vol_b"], 0, 4.0))
    want = z["harness/metrics"]                                                    # MSE, NMSE, SSIM, PSNR, maxval
    m = runner.metrics_to_dict(r.metric_vals["vol_a"][7])
    for k, w in zip(("MSE", "NMSE", "SSIM", "PSNR", "maxval"), want):
        assert abs(m[k] - float(w)) <= 2e-4 * max(1.0, abs(float(w))), (k, m[k], float(w))
    o, t = runner.postprocess(out[-1][-1], target)
    assert_close(o, T(z["harness/output"]), 1e-4, "abs / max image")
    assert_close(t, T(z["harness/target"]), 1e-6, "abs / max target")
    agg = r.test_epoch_end()
    assert agg["TotExamples"] == 2 and abs(agg["SSIM"] - float(want[2])) <= 2e-4 and abs(agg["PSNR"] - float(want[3])) <= 2e-3


def test_qcirim_128_filters_c32_256(dev):
    """(d) configs[4]: one qCIRIM cascade (8 steps, IndRNN 128 filters) at 4 echoes x 32 coils x 256 x 256 vs the oracle."""
    from mridc_amd.collections.quantitative.models.qcirim import qCIRIM
    cfg = {"quantitative_module_recurrent_layer": "IndRNN", "quantitative_module_conv_filters": [128, 128, 4],
           "quantitative_module_conv_kernels": [5, 3, 3], "quantitative_module_conv_dilations": [1, 2, 1],
           "quantitative_module_conv_bias": [True, True, False], "quantitative_module_recurrent_filters": [128, 128, 0],
           "quantitative_module_recurrent_kernels": [1, 1, 0], "quantitative_module_recurrent_dilations": [1, 1, 0],
           "quantitative_module_recurrent_bias": [True, True, False], "quantitative_module_depth": 2,
           "quantitative_module_time_steps": 8, "quantitative_module_num_cascades": 1, "quantitative_module_no_dc": True,
           "quantitative_module_signal_forward_model_sequence": "MEGRE", "quantitative_module_dimensionality": 2,
           "quantitative_module_gamma_regularization_factors": [150.0, 150.0, 1000.0, 150.0], "use_reconstruction_module": False,
           "fft_centered": False, "fft_normalization": "backward", "spatial_dims": [-2, -1], "coil_dim": 2,
           "coil_combination_method": "SENSE"}
    torch.manual_seed(0)
    model = qCIRIM(cfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rnn.ih.weight") or n.endswith("rnn.hh"):
                p.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    E, C, H, W = 4, 32, 256, 256
    TEs = [3.0, 11.5, 20.0, 28.5]
    g = torch.Generator().manual_seed(42)
    maps = [torch.rand(1, H, W, generator=g) * s for s in (60.0, 1.0, 30.0, 0.5)]         # R2* [1/s], S0, B0 [Hz], phi
    S = torch.randn(1, C, H, W, 2, generator=g) / C ** 0.5
    mask = (torch.rand(1, 1, 1, 1, W, 1, generator=g) < 0.3)
    y = torch.randn(1, E, C, H, W, 2, generator=g) * mask
    with torch.no_grad():
        ref = oracle.qrim.qcirim_forward(sd, cfg, maps[0], maps[1], maps[2], maps[3], TEs, y, S, None, mask)
    model = model.to(dev)
    with torch.no_grad():
        out = next(model(*[m_.to(dev) for m_ in maps], TEs, y.to(dev), S.to(dev), None, mask.to(dev)))
    for m_ in range(4):
        got = torch.stack([torch.stack(c) for c in out[1 + m_]])
        want = torch.stack([torch.stack(c) for c in ref[1 + m_]])
        assert_close(got, want, 5e-5, f"qCIRIM map {m_} (128 filters, 32 coils, 256 x 256)")


@pytest.mark.parametrize("chans,pools,pad", [(14, 2, 11), (18, 4, 15)], ids=["unet14x2", "unet18x4"])
def test_e2evn_cascade_full_size(dev, chans, pools, pad):
    """(e) configs[1]: one E2EVN cascade (VarNetBlock around NormUnet(14, 2, pad 11); and the (18, 4, pad 15) variant of the reference's
    default yaml) at 15 x 640 x 372 vs the oracle, in k-space and in the hybrid space the model runs its cascades in."""
    from mridc_amd.collections.reconstruction.models.unet_base.unet_block import NormUnet
    from mridc_amd.collections.reconstruction.models.varnet.vn_block import VarNetBlock
    torch.manual_seed(3)
    blk = VarNetBlock(NormUnet(chans, pools, padding_size=pad, normalize=True), fft_centered=False, fft_normalization="backward",
                      spatial_dims=[-2, -1], coil_dim=1, no_dc=False).eval()
    with torch.no_grad():
        blk.dc_weight.fill_(0.8)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    d = synthetic.make_slice(15, 640, 372, slice_idx=2, mask_dtype=torch.uint8)
    g = torch.Generator().manual_seed(9)
    pred = d["y"] + 0.05 * torch.randn(d["y"].shape, generator=g) * d["y"].abs().max()
    with torch.no_grad():
        ref = oracle.varnet.varnet_block_forward(sd, pred, d["y"], d["sensitivity_maps"], d["mask"], pools, pad, True, False, "backward",
                                                 [-2, -1], 1, False)
    blk = blk.to(dev)
    with torch.no_grad():
        got = blk(pred.to(dev), d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev))
    assert_close(got, ref, 5e-5, f"E2EVN cascade NormUnet({chans},{pools}) at 15 x 640 x 372")


@pytest.mark.parametrize("case", [(1, 15, 96, False, "backward"), (2, 7, 21, True, "ortho"), (1, 3, 9, True, "forward")],
                         ids=lambda c: f"B{c[0]}C{c[1]}H{c[2]}_{'c' if c[3] else 'n'}_{c[4]}")
def test_e2evn_chained_reduce_at_w372(dev, case):
    """Whole VarNet (3 cascades, NormUnet(8, 2)) at W = 372 in the hybrid space: with every block's data-consistency pass handing the next
    block its sens_reduce (mrx_pfa372_expand_reduce; the final SENSE combination is the last one) the output must be bit-identical to the
    unchained form (each block reducing the coil stack itself), and both must match the oracle (vn.py:94-142, vn_block.py:89-119)."""
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    B, C, H, centered, norm = case
    cfg = dict(num_cascades=3, channels=8, pooling_layers=2, padding_size=11, normalize=True, no_dc=False, use_sens_net=False,
               fft_centered=centered, fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE",
               train_loss_fn="l1", val_loss_fn="l1")
    torch.manual_seed(7)
    model = VarNet(cfg).eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("dc_weight"):
                p_.fill_(0.7)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    slices = [synthetic.make_slice(C, H, 372, slice_idx=3 + i) for i in range(B)]
    y = torch.cat([s_["y"] for s_ in slices], 0) * 50.0
    S = torch.cat([s_["sensitivity_maps"] for s_ in slices], 0)
    mask, target = slices[0]["mask"], torch.cat([s_["target"] for s_ in slices], 0)
    with torch.no_grad():
        ref = oracle.models.varnet_forward(sd, cfg, y, S, mask, None, target)
    model = model.to(dev)
    assert model._hybrid_ok(mask.to(dev))
    outs = {}
    keep = VarNet.chain_reduce
    try:
        for chain in (True, False):
            VarNet.chain_reduce = chain
            with torch.no_grad():
                outs[chain] = model(y.to(dev), S.to(dev), mask.to(dev), None, target.to(dev))
    finally:
        VarNet.chain_reduce = keep
    # (the chained pass and the separate reduction are two kernels: since the library dropped packed-fp32 instructions -- which pinned the
    # instruction sequence of the complex arithmetic -- the compiler may contract their multiply-adds differently: fp32 round-off, no more)
    assert rel_l2(torch.view_as_real(outs[True]), torch.view_as_real(outs[False])) <= 2e-6, "chained reduce changed the result"
    assert_close(torch.view_as_real(outs[True]), torch.view_as_real(ref), 5e-5, "VarNet 3 cascades at W = 372, hybrid space")


@pytest.mark.parametrize("sharing", [False, True], ids=["shared_block", "block_per_step"])
def test_recurrent_varnet_chained_reduce_at_w372(dev, sharing, monkeypatch):
    """Recurrent VarNet (4 steps, Conv2dGRU 12 channels x 3 layers, learned initializer) at W = 372 in the hybrid space: the k-space update of a
    step (recurrentvarnet.py:136-165) as ONE expand + update pass that also hands the next step its sens_reduce, against the unchained form
    (bit-identical) and the oracle (rvn.py:95-170)."""
    from mridc_amd.collections.reconstruction.models.rvn import RecurrentVarNet
    cfg = {"in_channels": 2, "recurrent_hidden_channels": 12, "recurrent_num_layers": 3, "num_steps": 4, "no_parameter_sharing": sharing,
           "learned_initializer": True, "initializer_initialization": "sense", "initializer_channels": [6, 6, 8, 8],
           "initializer_dilations": [1, 1, 2, 4], "initializer_multiscale": 1, "fft_centered": False, "fft_normalization": "backward",
           "spatial_dims": [-2, -1], "coil_dim": 1, "coil_combination_method": "SENSE", "pretrained": True}
    torch.manual_seed(13)
    model = RecurrentVarNet(cfg).eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    d = synthetic.make_slice(6, 24, 372, slice_idx=6)
    y = d["y"] * 50.0
    with torch.no_grad():
        ref = oracle.rvn.rvn_forward(sd, cfg, y, d["sensitivity_maps"], d["mask"], None, d["target"])
    model = model.to(dev)
    outs = {}
    for chain in ("1", "0"):
        monkeypatch.setenv("MRIDC_AMD_CHAIN_REDUCE", chain)
        with torch.no_grad():
            outs[chain] = model(y.to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev), None, d["target"].to(dev))
    assert rel_l2(torch.view_as_real(outs["1"]), torch.view_as_real(outs["0"])) <= 2e-6, "chained reduce changed the result"   # (fp32 round-off: see above)
    assert_close(torch.view_as_real(outs["1"]), torch.view_as_real(ref), 5e-5, "Recurrent VarNet at W = 372, hybrid space")


@pytest.mark.parametrize("no_dc", [True, False], ids=["no_dc_model_zoo", "with_dc"])
def test_cascadenet_chained_reduce_at_w372(dev, no_dc):
    """CascadeNet (3 cascades x 3 convs, 16 channels) at W = 372 in the hybrid space, chained (each block's expand pass hands the next block its
    sens_reduce: mrx_pfa372_expand_reduce with and without the data-consistency epilogue) against unchained: bit-identical; and against the
    oracle (ccnn.py:93-142, ccnn_block.py:101-139)."""
    from mridc_amd.collections.reconstruction.models.ccnn import CascadeNet
    cfg = dict(num_cascades=3, hidden_channels=16, n_convs=3, batchnorm=False, no_dc=no_dc, use_sens_net=False, fft_centered=False,
               fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE", train_loss_fn="l1",
               val_loss_fn="l1")
    torch.manual_seed(11)
    model = CascadeNet(cfg).eval()
    # (.cpu(): the default `activation=nn.PReLU()` argument is ONE module shared by every Conv2d ever built, as in the reference, conv2d.py:14 --
    #  an earlier test may have moved it to the GPU)
    sd = {k: v.detach().cpu().clone() for k, v in model.cpu().state_dict().items()}
    d = synthetic.make_slice(7, 40, 372, slice_idx=4)
    y = d["y"] * 50.0
    with torch.no_grad():
        ref = oracle.cascadenet.cascadenet_forward(sd, cfg, y, d["sensitivity_maps"], d["mask"], None, d["target"])
    model = model.to(dev)
    outs = {}
    keep = CascadeNet.chain_reduce
    try:
        for chain in (True, False):
            CascadeNet.chain_reduce = chain
            with torch.no_grad():
                outs[chain] = model(y.to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev), None, d["target"].to(dev))
    finally:
        CascadeNet.chain_reduce = keep
    # (the chained pass and the separate reduction are two kernels: since the library dropped packed-fp32 instructions -- which pinned the
    # instruction sequence of the complex arithmetic -- the compiler may contract their multiply-adds differently: fp32 round-off, no more)
    assert rel_l2(torch.view_as_real(outs[True]), torch.view_as_real(outs[False])) <= 2e-6, "chained reduce changed the result"
    assert_close(torch.view_as_real(outs[True]), torch.view_as_real(ref), 5e-5, "CascadeNet at W = 372, hybrid space")


@pytest.mark.parametrize("case", [(1, 15, 640, True, "ortho"), (2, 7, 21, False, "backward"), (1, 3, 9, True, "forward")],
                         ids=lambda c: f"B{c[0]}C{c[1]}H{c[2]}_{'c' if c[3] else 'n'}_{c[4]}")
def test_pfa372_row_operators_and_general_mask_gradient(dev, case):
    """The W = 372 prime-factor row operators against the oracle: the hybrid-space sens_expand / sens_reduce / expand + soft DC of the
    E2EVN cascades (vn_block.py:51-119 with k-space kept as IFFT_H(k)), and log_likelihood_gradient for a 2-D (row-dependent) mask, whose
    row passes run on the same kernels around the column pass (rim_utils.py:11-67)."""
    from mridc_amd import ops
    B, C, H, centered, norm = case
    W = 372
    y, S, mask1d, eta = _problem(B, C, H, W, 7000 + C, centered, norm)
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, H, W, 2, generator=g)
    yd, Sd, ed, xd = y.to(dev), S.to(dev), eta.to(dev), x.to(dev)
    # the W-only transforms: FFT_W(x S) and sum_c conj(S) IFFT_W(k), defined through the full transforms of the oracle
    full = oracle.fft.fft2(oracle.utils.complex_mul(x.unsqueeze(1), S), centered, norm)              # fft2(x S)
    yt_full = ops.llg_prepare(full.to(dev), centered, norm)                                           # IFFT_H of it = FFT_W(x S)
    got = ops.sens_expand(xd, Sd, centered, norm, hybrid=True)
    assert_close(got, yt_full, 1e-5, "sens_expand (W transform only)")
    kh = ops.llg_prepare(yd, centered, norm)                                                          # hybrid-space data
    want_red = oracle.varnet.sens_reduce(y, S, centered, norm, [-2, -1], 1).squeeze(1)
    assert_close(ops.sens_reduce(kh, Sd, centered, norm, hybrid=True), want_red, 1e-5, "sens_reduce from hybrid space")
    w = torch.tensor([0.7])
    pred = yd + 0.1 * torch.randn(yd.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    m8 = mask1d.to(torch.uint8).to(dev)
    predh, refh = ops.llg_prepare(pred, centered, norm), kh
    got = ops.sens_expand_dc_hybrid(xd, Sd, predh, refh, m8, w.to(dev), centered, norm)
    want = predh - torch.where(m8.bool(), predh - refh, torch.zeros(1, device=dev)) * 0.7 - yt_full
    assert_close(got, want, 1e-5, "expand + soft DC in hybrid space")
    # general (row-dependent) mask
    m2d = (torch.rand(1, 1, H, W, 1, generator=g) < 0.3)
    y2 = full * m2d + 0.05 * y
    with torch.no_grad():
        g_ref = oracle.rim.log_likelihood_gradient(eta, y2, S, m2d, 1.2, centered, norm, [-2, -1], 1).contiguous()
    assert_close(ops.llg(ed, y2.to(dev), Sd, m2d.to(dev), 1.2, centered, norm), g_ref, 1e-5, "log_likelihood_gradient, 2-D mask, W = 372")


@pytest.mark.parametrize("case", [(1, 15, 640, True, "ortho", torch.bool, False), (2, 7, 21, False, "backward", torch.float32, True),
                                  (1, 3, 9, True, "forward", torch.uint8, False), (1, 32, 320, False, "none", torch.bool, False)],
                         ids=lambda c: f"B{c[0]}C{c[1]}H{c[2]}_{'c' if c[3] else 'n'}_{c[4]}")
def test_general_mask_gradient_column_tiled_at_w372(dev, case):
    """log_likelihood_gradient (rim_utils.py:11-67) for row-dependent masks at W = 372 on the column-tiled coil stack (mrx_tile4_cols,
    mrx_pfa372_expand_t4, mrx_llg_cols_dc_t4, mrx_pfa372_reduce_t4): vs the oracle, bit-identical to the row-major three-pass form, and
    the deferred form (coil-group partial sums read by layer 1's tile loader) vs oracle conv5x5 + IndRNN."""
    from mridc_amd import ops
    B, C, H, centered, norm, mdt, batched = case
    W = 372
    y, S, _, eta = _problem(B, C, H, W, 9000 + C + H, centered, norm)
    g = torch.Generator().manual_seed(C + H)
    m2d = (torch.rand(B if batched else 1, 1, H, W, 1, generator=g) < 0.35)
    full = oracle.fft.fft2(oracle.utils.complex_mul(torch.randn(B, 1, H, W, 2, generator=g), S), centered, norm)
    y2 = (full / full.abs().max()) * m2d + 0.05 * y
    mask = m2d if mdt == torch.bool else m2d.to(mdt)
    sigma = 0.9
    w, b, wi, bi, hh = _layer1(78)
    h_prev = torch.randn(B, 64, H, W, generator=torch.Generator().manual_seed(6)).relu()
    with torch.no_grad():
        g_ref = oracle.rim.log_likelihood_gradient(eta, y2, S, mask, sigma, centered, norm, [-2, -1], 1).contiguous()
        h_ref = oracle.rim.indrnn_cell(oracle.rim.conv_nonlinear(g_ref, w, b, 5, 1, "relu"), h_prev, wi, bi, hh, 1, 1)
    yd, Sd, md, ed, hp = y2.to(dev), S.to(dev), mask.to(dev), eta.to(dev), h_prev.to(dev)
    assert ops.llg_t4_supported(yd)
    got = ops.llg(ed, yd, Sd, md, sigma, centered, norm)
    assert_close(got, g_ref, 1e-5, "column-tiled general-mask gradient vs oracle")
    keep = ops.LLG_T4
    try:
        ops.LLG_T4 = False
        assert not ops.llg_t4_supported(yd)
        row_major = ops.llg(ed, yd, Sd, md, sigma, centered, norm)
    finally:
        ops.LLG_T4 = keep
    assert torch.equal(got, row_major), "the tiled layout changed the arithmetic"
    # the measured data tiled once: [B*C][93][H][4]
    t4 = ops._y_t4(yd)
    want_t4 = yd.reshape(B * C, H, 93, 4, 2).permute(0, 2, 1, 3, 4).contiguous()
    assert torch.equal(t4.reshape(-1), want_t4.reshape(-1))
    # deferred form: the coil-group partial sums of the last pass + (round 4) ONE constant plane -A^H M y instead of y in every column pass
    wd, bd, wid, bid, hhd = (t.to(dev) for t in (w, b, wi, bi, hh))
    keep_noy = ops.LLG_T4_NO_Y
    try:
        for no_y in (True, False):
            ops.LLG_T4_NO_Y = no_y
            part, n = ops.llg(ed, yd, Sd, md, sigma, centered, norm, parts=True)
            assert n == -(-C // 5) + (1 if no_y else 0)
            if no_y:
                total = part[:n].sum(0) / sigma ** 2          # the planes add up to the gradient's last two channels
                assert_close(total, g_ref[:, 2:4].permute(0, 2, 3, 1), 1e-5, "partial planes incl. the constant one vs oracle")
                again, n2 = ops.llg(ed, yd, Sd, md, sigma, centered, norm, parts=True)
                assert again.data_ptr() == part.data_ptr() and n2 == n                # one buffer per slice: the constant plane is made once
            h_def = ops.rim_layer_indrnn_packed_llg(ed, part, n, sigma, ops.rim_layer_pack(wd, wid), 64, 5, 1, bd, bid, hhd, hp)
            assert_close(h_def, h_ref, 1e-5, f"layer 1 reading the general-mask partial sums vs oracle (constant plane: {no_y})")
    finally:
        ops.LLG_T4_NO_Y = keep_noy


def test_rim_block_general_mask_at_w372(dev):
    """One RIMBlock (8 steps, IndRNN 64) at 1 x 15 x 640 x 372 with a 2-D (row-dependent) random mask vs the oracle: the deferred
    column-tiled gradient route of RIMBlock.forward, and the same block with the tiled path switched off (row-major three-pass form)."""
    from mridc_amd import ops
    cfg, model, sd = _cirim(dict(num_cascades=1), 1.0)
    d = synthetic.make_slice(15, 640, 372, slice_idx=4)
    g = torch.Generator().manual_seed(11)
    m2d = torch.rand(1, 1, 640, 372, 1, generator=g) < 0.3
    m2d[:, :, 300:340, 170:202] = True
    y = d["kspace"] * m2d
    rc = oracle.rim.RIMConfig(**{k: cfg[k] for k in ("recurrent_layer", "conv_filters", "conv_kernels", "conv_dilations", "conv_bias",
                                                     "recurrent_filters", "recurrent_kernels", "recurrent_dilations", "recurrent_bias",
                                                     "depth", "no_dc", "fft_centered", "fft_normalization", "spatial_dims", "coil_dim")},
                              time_steps=8)
    p = {k[len("cirim.0."):]: v for k, v in sd.items() if k.startswith("cirim.0.")}
    with torch.no_grad():
        ref, ref_hx = oracle.rim.rim_block_forward(p, rc, y, y, d["sensitivity_maps"], m2d, None, None, 1.0, False)
    blk = model.cirim[0].to(dev)
    yd, S, m = y.to(dev), d["sensitivity_maps"].to(dev), m2d.to(dev)
    assert not ops.mask_is_row_invariant(m)
    outs = {}
    keep = ops.LLG_T4
    try:
        for t4 in (True, False):
            ops.LLG_T4 = t4
            with torch.no_grad():
                etas, hx = blk(yd, yd, S, m, None, None, 1.0, keep_eta=False)
            assert_close(torch.stack(etas), torch.stack(ref), 2e-5, f"8-step block, 2-D mask, tiled={t4}")
            for j in range(2):
                assert_close(hx[j], ref_hx[j], 2e-5, f"hidden state {j}, tiled={t4}")
            outs[t4] = etas[-1]
    finally:
        ops.LLG_T4 = keep
    assert rel_l2(outs[True], outs[False]) <= 5e-6


def test_general_mask_gradient_with_the_tap_gather_folded_in(dev):
    """Round 5: for general (2-D) masks at W = 372 the nine-tap gather that ends a RIM step rides in the FIRST pass of the next step's gradient
    (mrx_pfa372_expand_t4_gather).  The operator: eta_new bit-identical to mrx_rim_final_gather, partial planes bit-identical to the gradient of that
    eta.  The block: 8 steps with the fold on and off give bit-identical estimates (15 x 640 x 372 and a ragged small shape)."""
    from mridc_amd import ops
    g = torch.Generator().manual_seed(23)
    for B, C, H in ((2, 15, 640), (1, 6, 37)):
        W = 372
        r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
        eta, taps, bfin = r(B, H, W, 2), r(B, 18, H, W) * 0.1, r(2) * 0.1
        S, y = r(B, C, H, W, 2) * 0.3, r(B, C, H, W, 2)
        m = (torch.rand(B, 1, H, W, 1, generator=g) < 0.3).to(dev)
        y = y * m
        want_eta = ops.rim_final_gather(taps, bfin, eta)
        part0, n0 = ops.llg(want_eta, y, S, m, 1.0, False, "backward", parts=True)
        part0 = part0[:n0].clone()
        part1, n1, got_eta = ops.llg(eta, y, S, m, 1.0, False, "backward", parts=True, gather=(taps, bfin))
        assert n1 == n0 and torch.equal(got_eta, want_eta) and torch.equal(part1[:n1], part0)
    cfg, model, sd = _cirim(dict(num_cascades=1), 1.0)
    d = synthetic.make_slice(15, 640, 372, slice_idx=4)
    m2d = torch.rand(1, 1, 640, 372, 1, generator=torch.Generator().manual_seed(11)) < 0.3
    m2d[:, :, 300:340, 170:202] = True
    blk = model.cirim[0].to(dev)
    yd, S, m = (d["kspace"] * m2d).to(dev), d["sensitivity_maps"].to(dev), m2d.to(dev)
    outs = {}
    keep, keep_q = ops.LLG_T4_GATHER, ops.RIM_TAPS_Q
    try:
        ops.RIM_TAPS_Q = False                 # (the fold reads the eighteen tap planes; the unfolded default route pre-sums them: another order of additions)
        for fold in (True, False):
            ops.LLG_T4_GATHER = fold
            calls = []
            orig = ops.rim_final_gather
            ops.rim_final_gather = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
            try:
                with torch.no_grad():
                    etas, _ = blk(yd, yd, S, m, None, None, 1.0, keep_eta=False)
            finally:
                ops.rim_final_gather = orig
            assert len(calls) == (1 if fold else blk.time_steps)            # folded: only the last step's gather is its own launch
            outs[fold] = torch.stack(etas)
    finally:
        ops.LLG_T4_GATHER, ops.RIM_TAPS_Q = keep, keep_q
    assert torch.equal(outs[True], outs[False])


@pytest.mark.parametrize("shape", [(15, 640, 372), (6, 37, 75)])
def test_rim_block_fp16_route_on_and_off(dev, shape):
    """RIMBlock with the dominant layer's convolution on two-term fp16 operands (the default: stack 0 keeps the bound of its outputs, stack 1
    scales by it) and on the three-term bf16 form (MRIDC_AMD_ARITH=bf16x3), 8 steps, both against the oracle and against each other."""
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    C, H, W = shape
    cfg, model, sd = _cirim(dict(num_cascades=1), 1.0)
    d = synthetic.make_slice(C, H, W, slice_idx=5)
    rc = oracle.rim.RIMConfig(**{k: cfg[k] for k in ("recurrent_layer", "conv_filters", "conv_kernels", "conv_dilations", "conv_bias",
                                                     "recurrent_filters", "recurrent_kernels", "recurrent_dilations", "recurrent_bias",
                                                     "depth", "no_dc", "fft_centered", "fft_normalization", "spatial_dims", "coil_dim")},
                              time_steps=8)
    p = {k[len("cirim.0."):]: v for k, v in sd.items() if k.startswith("cirim.0.")}
    with torch.no_grad():
        ref, ref_hx = oracle.rim.rim_block_forward(p, rc, d["y"], d["y"], d["sensitivity_maps"], d["mask"], None, None, 1.0, False)
    blk = model.cirim[0].to(dev)
    y, S, m = d["y"].to(dev), d["sensitivity_maps"].to(dev), d["mask"].to(dev)
    assert blk._f16_route()
    outs = {}
    keep = RIMBlock.layer2_f16
    try:
        for f16 in (True, False):
            RIMBlock.layer2_f16 = f16
            with torch.no_grad():
                etas, hx = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
            assert_close(torch.stack(etas), torch.stack(ref), 2e-5, f"8-step block, fp16 route {f16}")
            for j in range(2):
                assert_close(hx[j], ref_hx[j], 2e-5, f"hidden state {j}, fp16 route {f16}")
            outs[f16] = etas[-1]
    finally:
        RIMBlock.layer2_f16 = keep
    assert rel_l2(outs[True], outs[False]) <= 5e-6


@pytest.mark.parametrize("weights", ["bench", "boosted"])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_one_cascade_training_at_headline_size(dev, precision, weights):
    """BASELINE config 4 at the headline shape (base_cirim_train.yaml:175-180): ONE cascade (8 time-steps) of the CIRIM at 1 x 15 x 640 x 372,
    forward + l1 loss + backward on the explicit tape (`training.cirim_forward_backward`) against torch autograd of the oracle -- the loss and
    all 11 parameter gradients -- on TWO sets of weights: the bench's (seed 0, reference initialisation: what `bench.py --train` reports its
    parity on) and a boosted set (seed 5, biases 0.05, recurrent weights x 3: the ReLUs of the recurrence bite).
    fp32: every gradient rel-L2 <= 2e-3.  bf16 (the reference's `precision: 16`): the whole gradient vector against each arithmetic of
    oracle/amp.py within tests/_util.py TRAIN_TOL (bench weights) / TRAIN_TOL_BOOSTED -- the kernels' own arithmetic restated on the CPU (tight: a
    kernel bug shows here), torch.autocast (the reference's semantics) and fp32 (what bf16 costs; the autocast oracle itself sits 1e-2 .. 8e-2 from
    it) -- AND each of the 11 gradients on its own against the first two (TRAIN_TOL_PER_TENSOR): `rnn.hh` has |g| ~ 4e-4 of the vector norm, a
    completely wrong hh gradient moves the whole-vector figure by 2e-4."""
    from mridc_amd import autograd as ag
    from mridc_amd import training
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    from tests._util import TRAIN_TOL, TRAIN_TOL_BOOSTED, TRAIN_TOL_PER_TENSOR
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=1)
    seed, sl = (0, 0) if weights == "bench" else (5, 7)
    torch.manual_seed(seed)
    model = CIRIM(cfg)
    if weights == "boosted":
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("bias"):
                    p_.normal_(0, 0.05)
                if n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh"):
                    p_.mul_(3.0)                           # the reference init is nearly linear: make the ReLUs of the recurrence bite
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s = synthetic.make_slice(15, 640, 372, slice_idx=sl)
    refs = {"fp32": oracle.amp.cirim_loss_and_gradients(state, cfg, s, "fp32")}
    if precision == "bf16":
        refs["autocast_bf16"] = oracle.amp.cirim_loss_and_gradients(state, cfg, s, "autocast_bf16")
        emul = dict(round_results=True) if training.BF16_STORAGE else dict(fp32_forward=((64, 2),))
        refs["kernel_arithmetic"] = oracle.amp.cirim_loss_and_gradients(state, cfg, s, "bf16_operands", **emul)
    model = model.to(dev).train()
    batch = {k: s[k].to(dev) for k in ("y", "sensitivity_maps", "mask", "target")}
    assert training._tape_supported(model, batch)
    ag.set_precision(precision)
    try:
        for prm in model.parameters():
            prm.grad = None
        loss = training.cirim_forward_backward(model, batch, precision)
    finally:
        ag.set_precision("f32")
    names = [n_ for n_, _ in model.named_parameters() if not n_.endswith("dc_weight")]
    assert len(names) == 11
    grads = dict(model.named_parameters())
    got_all = torch.cat([grads[n_].grad.detach().cpu().reshape(-1).double() for n_ in names])
    errs = {}
    for m_, (ref_loss, ref_g) in refs.items():
        want = torch.cat([ref_g[n_].reshape(-1).double() for n_ in names])
        errs[m_] = (float((got_all - want).norm() / want.norm()), abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    print(f"training parity at 15 x 640 x 372, {precision}, {weights} weights (whole gradient, loss):", errs)
    if precision == "f32":
        assert errs["fp32"][1] <= 1e-5, errs
        for n_ in names:
            assert_close(grads[n_].grad, refs["fp32"][1][n_], 2e-3, f"gradient of {n_} at 15 x 640 x 372")
    for m_, tol in (TRAIN_TOL if weights == "bench" else TRAIN_TOL_BOOSTED)[precision].items():
        assert errs[m_][0] <= tol and errs[m_][1] <= 2e-2, (m_, errs)
    if precision == "bf16":
        per = {m_: {n_: float((grads[n_].grad.detach().cpu().reshape(-1).double() - refs[m_][1][n_].reshape(-1).double()).norm()
                              / refs[m_][1][n_].reshape(-1).double().norm()) for n_ in names} for m_ in TRAIN_TOL_PER_TENSOR[weights]}
        print("per tensor:", {m_: {n_.replace("cirim.0.", ""): f"{e_:.2e}" for n_, e_ in d_.items()} for m_, d_ in per.items()})
        for m_, tol in TRAIN_TOL_PER_TENSOR[weights].items():
            worst = max(per[m_], key=per[m_].get)
            assert per[m_][worst] <= tol, (m_, worst, per[m_][worst], tol)
