"""GPU parity of the precision-16 inference route of the U-Net models and of the wide plain 3x3 convolutions (qCIRIM) (`trainer.precision: 16`, reference base_vn_run.yaml:98 / base_unet_run.yaml:96 = native AMP,
torch.autocast(float16) around the forward pass): the 3x3 convolutions of unet_block.py:250-259 on ONE fp16 term (mrx_unet_conv3x3_p16, csrc/unet_f16.hip).
Checkers, all on the CPU: the kernel's arithmetic restated (oracle.amp.fp16_kernel_arithmetic: fp16-rounded operands, wide sums -- tight), the reference's own
arithmetic (oracle.amp.autocast_fp16: what the reference computes -- SURVEY appendix C's 3e-2), and the fp32 oracle (the route must sit no further from it
than autocast does)."""
import pytest
import torch
import torch.nn.functional as Fn

import oracle
from mridc_amd import synthetic
from tests._util import rel_l2

pytestmark = pytest.mark.gpu

OP_TOL = 2e-6            # operator against float64 sums of the fp16-rounded operands (plain sources: the same roundings exactly)
OP_TOL_LAZY = 3e-5       # lazy sources: the kernel normalises in fp32 before it rounds, the checker in float64 -- a value in 1e4 rounds the other way
NET_TOL_KERNEL = 2e-3    # one NormUnet pass against the restated kernel arithmetic (measured 6e-4 .. 1.1e-3: rounding flips, see the test)
NET_TOL_AUTOCAST = 3e-2  # against the reference under autocast(float16)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _r16(t):
    return oracle.amp.fp16_round(t.float()).double()


@pytest.mark.parametrize("shape", [(1, 2, 0, 14, 64, 40), (2, 14, 14, 14, 37, 75), (1, 28, 28, 28, 160, 95), (3, 5, 3, 36, 9, 7), (1, 56, 0, 56, 33, 47),
                                   (8, 14, 0, 14, 200, 700), (8, 14, 14, 14, 640, 372)], ids=lambda s: "x".join(map(str, s)))
def test_unet_conv3x3_precision16_is_fp16_operands_with_wide_sums(shape, dev):
    """mrx_unet_conv3x3_p16 (every cout-block count, 8- and 16-row work items, one and two sources, plain and lazy) against the CPU float64 convolution of the
    fp16-rounded operands; the statistics against those of that result; and the fp32-class route untouched outside the context."""
    from mridc_amd import ops
    B, Ca, Cb, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    a_raw, b_raw = r(B, Ca, H, W) * 2 + 0.5, (r(B, Cb, H, W) - 0.3 if Cb else None)
    w = r(Cout, Ca + Cb, 3, 3) / (9 * (Ca + Cb)) ** 0.5

    def stats(t):
        return torch.stack([t.mean((2, 3)), 1.0 / torch.sqrt(t.var((2, 3), unbiased=False) + 1e-5)], -1)

    na, nb = stats(a_raw), (stats(b_raw) if Cb else None)

    def lazy64(raw, n):
        return Fn.leaky_relu((raw.double() - n[..., 0, None, None].double()) * n[..., 1, None, None].double(), 0.2)

    for lazy in (False, True):
        xs = [lazy64(a_raw, na) if lazy else a_raw.double()]
        if Cb:
            xs.append(lazy64(b_raw, nb) if lazy else b_raw.double())
        ref = Fn.conv2d(_r16(torch.cat(xs, 1)), _r16(w), padding=1)                       # CPU, float64
        src_a = (a_raw.to(dev), na.to(dev)) if lazy else a_raw.to(dev)
        src_b = None if not Cb else ((b_raw.to(dev), nb.to(dev)) if lazy else b_raw.to(dev))
        with ops.inference_precision(16):
            y, norm = ops.unet_conv3x3(src_a, src_b, w.to(dev))
        tol = OP_TOL_LAZY if lazy else OP_TOL
        assert rel_l2(y, ref) <= tol, (shape, lazy, rel_l2(y, ref))
        mean, var = ref.mean((2, 3)), ref.var((2, 3), unbiased=False)
        assert (norm[..., 0].double().cpu() - mean).abs().max() <= tol * max(1.0, float(ref.abs().max()))
        assert rel_l2(norm[..., 1], 1.0 / torch.sqrt(var + 1e-5)) <= 10 * tol
        full = Fn.conv2d(torch.cat(xs, 1), w.double(), padding=1)
        y32, _ = ops.unet_conv3x3(src_a, src_b, w.to(dev))                                # outside the context: fp32-class
        assert rel_l2(y32, full) <= 2e-6
        assert 1e-5 <= rel_l2(y, full) <= 2e-3                                            # ... and the one-term result is fp16-operand-class, not more, not less


def _load(model, sd, dev):
    model.load_state_dict(sd, strict=False)
    return model.to(dev).eval()


@pytest.mark.parametrize("cfg", [(14, 2, 11, 640, 372), (18, 4, 15, 160, 96), (8, 3, 7, 45, 37)], ids=lambda c: f"{c[0]}ch_{c[1]}pools_{c[3]}x{c[4]}")
def test_norm_unet_precision16_sits_at_the_restated_kernel_arithmetic(cfg, dev):
    """ONE NormUnet pass (ten to eighteen 3x3 convolutions, unet_block.py:139-308) inside `inference_precision(16)` against the oracle with the kernels' arithmetic
    restated.  Operator by operator the kernel makes the checker's roundings exactly (the test above; 8e-8 when the checker normalises in fp32 like the kernel,
    profiles/r06_unet_p16_error_sources.txt); through a network the two InstanceNorm statistics differ by ~1e-6 (torch's fp32 sums against the kernels' tile sums
    merged in double), one operand in ~500 then rounds to the other fp16 neighbour, and ten layers carry that to 6e-4 .. 1.1e-3 -- measured, about half the
    1.4e-3 .. 2e-3 that fp16 operands cost against fp32.  So: within NET_TOL_KERNEL of the restated arithmetic, clearly closer to it than to fp32, and at the
    fp32 oracle's distance from it that the restated arithmetic itself has."""
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.unet_base.unet_block import NormUnet
    chans, pools, pad, H, W = cfg
    torch.manual_seed(chans + pools)
    net = NormUnet(chans, pools, padding_size=pad).eval()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(1, 1, H, W, 2, generator=torch.Generator().manual_seed(H))
    with torch.no_grad():
        ref32 = oracle.unet.norm_unet_forward(sd, x, pools, padding_size=pad)
        with oracle.amp.fp16_kernel_arithmetic():
            refk = oracle.unet.norm_unet_forward(sd, x, pools, padding_size=pad)
        net = net.to(dev)
        with ops.inference_precision(16):
            got = net(x.to(dev))
        got32 = net(x.to(dev))
    e_k, e_32, k_32 = rel_l2(got, refk), rel_l2(got, ref32), rel_l2(refk, ref32)
    assert rel_l2(got32, ref32) <= 2e-5
    assert e_k <= NET_TOL_KERNEL and e_k <= 0.8 * e_32, (e_k, e_32)
    assert 0.7 * k_32 <= e_32 <= 1.4 * k_32 and 1e-4 <= e_32 <= 1e-2, (e_32, k_32)


@pytest.mark.parametrize("case", [(4, 48, 40, 2, 0), (15, 160, 372, 6, 1), (3, 64, 372, 3, 2)], ids=lambda c: f"C{c[0]}_{c[1]}x{c[2]}_{c[3]}casc")
def test_varnet_precision16_against_kernel_arithmetic_autocast_and_fp32(case, dev):
    """VarNet (E2EVN, base_vn_run.yaml) with `precision: 16` from its cfg on the HIP path -- hybrid cascades at W = 372, the k-space form elsewhere --
    against the three CPU checkers."""
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    C, H, W, ncasc, seed = case
    cfg = dict(synthetic.E2EVN_BASELINE_CFG, num_cascades=ncasc)
    torch.manual_seed(seed)
    model = VarNet(dict(cfg, precision=16)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    d = synthetic.make_slice(C=C, H=H, W=W, slice_idx=seed)
    args = (d["y"], d["sensitivity_maps"], d["mask"], None, d["target"])
    with torch.no_grad():
        ref32 = oracle.models.varnet_forward(sd, cfg, *args)
        with oracle.amp.fp16_kernel_arithmetic():
            refk = oracle.models.varnet_forward(sd, cfg, *args)
        with oracle.amp.autocast_fp16():
            refa = oracle.models.varnet_forward(sd, cfg, *args)
    model = model.to(dev)
    with torch.no_grad():
        got = model(*(None if a is None else a.to(dev) for a in args))
        model.precision = 32
        got32 = model(*(None if a is None else a.to(dev) for a in args))
    v = lambda t: torch.view_as_real(t.to(torch.complex64)).double().cpu()  # noqa: E731
    e_k, e_a, e_32, a_32 = rel_l2(v(got), v(refk)), rel_l2(v(got), v(refa)), rel_l2(v(got), v(ref32)), rel_l2(v(refa), v(ref32))
    assert rel_l2(v(got32), v(ref32)) <= 1e-4
    assert e_a <= NET_TOL_AUTOCAST, (case, e_a)                          # the stated tolerance against the reference's own arithmetic
    assert e_k <= max(e_32, 2e-4), (case, e_k, e_32)                     # (closer to its restated arithmetic than to fp32; tight per network pass: the test above)
    assert e_32 <= max(2.0 * a_32, 2e-4), (case, e_32, a_32)              # no further from fp32 than the reference's own precision-16 arithmetic
    assert e_32 >= 1e-6                                                   # (and it IS the one-term route)


def test_unet_model_precision16_from_the_trainer(dev):
    """UNet (base_unet_run.yaml:96) takes its precision from `trainer.precision` as the reference hands it to pytorch-lightning; MRIDC_AMD_PRECISION is the
    process default; gradients recorded (training) keep the fp32-class convolutions."""
    import types
    from mridc_amd.collections.reconstruction.models.unet import UNet
    cfg = dict(synthetic.E2EVN_BASELINE_CFG)
    torch.manual_seed(3)
    m16 = UNet(cfg, trainer=types.SimpleNamespace(precision=16)).eval()
    sd = {k: v.detach().clone() for k, v in m16.state_dict().items()}
    m32 = _load(UNet(cfg), sd, dev)
    m16 = m16.to(dev)
    d = synthetic.make_slice(C=4, H=64, W=48, slice_idx=1)
    args = (d["y"], d["sensitivity_maps"], d["mask"], None, d["target"])
    with torch.no_grad():
        with oracle.amp.fp16_kernel_arithmetic():
            refk = oracle.models.unet_model_forward(sd, cfg, *args)
        ref32 = oracle.models.unet_model_forward(sd, cfg, *args)
        g16 = m16(*(None if a is None else a.to(dev) for a in args))
        g32 = m32(*(None if a is None else a.to(dev) for a in args))
    v = lambda t: torch.view_as_real(t.to(torch.complex64)).double().cpu()  # noqa: E731
    assert rel_l2(v(g16), v(refk)) <= NET_TOL_KERNEL
    assert rel_l2(v(g32), v(ref32)) <= 1e-4
    assert rel_l2(v(g16), v(g32)) >= 1e-6


# (B, Cin, Cout, H, W, dilation, pad mode (0 zero, 1 replicate), activation (0 none, 1 ReLU, 2 LeakyReLU))
@pytest.mark.parametrize("case", [(1, 128, 128, 64, 64, 2, 1, 1), (2, 64, 32, 21, 40, 1, 0, 2), (1, 24, 70, 33, 47, 2, 1, 0), (1, 128, 128, 256, 256, 2, 1, 1)],
                         ids=lambda c: f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_d{c[5]}")
def test_conv3x3_precision16_is_fp16_operands_with_wide_sums(case, dev):
    """mrx_conv3x3_p16 (the plain 3x3 convolution of the wide layers -- the qRIM's 128 -> 128 dilation 2, conv_layers.py:121-123; bias + activation epilogue, zero and
    replicate padding, every cout-block count) against the CPU float64 convolution of the fp16-rounded operands; fp32-class outside the context."""
    from mridc_amd import ops
    B, Cin, Cout, H, W, dil, pad_mode, act = case[0], case[1], case[2], case[3], case[4], case[5], case[6], case[7]
    g = torch.Generator().manual_seed(Cin + H)
    x, w, b = torch.randn(B, Cin, H, W, generator=g), torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, torch.randn(Cout, generator=g)

    def ref_of(xx, ww):
        xp = Fn.pad(xx, (dil,) * 4, mode="replicate" if pad_mode == ops.PAD_REPLICATE else "constant")
        y = Fn.conv2d(xp, ww, b.double(), dilation=dil)
        return Fn.relu(y) if act == ops.ACT_RELU else (Fn.leaky_relu(y, 0.1) if act == ops.ACT_LEAKY else y)

    with ops.inference_precision(16):
        got = ops.conv3x3_h(x.to(dev), w.to(dev), b.to(dev), dil, pad_mode, act, 0.1)
    got32 = ops.conv3x3_h(x.to(dev), w.to(dev), b.to(dev), dil, pad_mode, act, 0.1)
    assert rel_l2(got, ref_of(_r16(x), _r16(w))) <= OP_TOL, rel_l2(got, ref_of(_r16(x), _r16(w)))
    assert rel_l2(got32, ref_of(x.double(), w.double())) <= 2e-6
    assert 1e-5 <= rel_l2(got, got32) <= 2e-3


@pytest.mark.parametrize("case", [(1, 64, 64, True), (2, 21, 40, False), (1, 256, 256, True)], ids=lambda c: f"B{c[0]}_{c[1]}x{c[2]}_{'cell' if c[3] else 'plain'}")
def test_conv1x1_128_precision16_is_fp16_operands_with_wide_sums(case, dev):
    """mrx_conv1x1_sq_p16 (the 128 -> 128 1x1 convolution with the IndRNN cell as its epilogue, rnn_cells.py:384-391: the qRIM's cells): relu(W16 x16 + b + hh * h_prev)
    with x and W rounded to fp16 once and everything else in fp32, against the CPU in float64; the bound it attaches for a following 3x3 layer; fp32-class outside."""
    from mridc_amd import ops
    B, H, W, cell = case
    g = torch.Generator().manual_seed(H + W)
    x, w, b = torch.randn(B, 128, H, W, generator=g) * 3, torch.randn(128, 128, 1, 1, generator=g) / 128 ** 0.5, torch.randn(128, generator=g)
    hh, hp = (torch.rand(128, generator=g), torch.randn(B, 128, H, W, generator=g)) if cell else (None, None)

    def ref_of(xx, ww):
        y = Fn.conv2d(xx, ww, b.double())
        if cell:
            y = y + hh.double().view(1, -1, 1, 1) * hp.double()
        return Fn.relu(y)

    args = (x.to(dev), w.to(dev), b.to(dev), ops.ACT_RELU, 0.0, hh.to(dev) if cell else None, hp.to(dev) if cell else None)
    with ops.inference_precision(16):
        got = ops.conv1x1_64(*args)
    got32 = ops.conv1x1_64(*args)
    want = ref_of(_r16(x), _r16(w))
    assert rel_l2(got, want) <= OP_TOL, rel_l2(got, want)
    assert rel_l2(got32, ref_of(x.double(), w.double())) <= 2e-6
    assert 1e-5 <= rel_l2(got, got32) <= 2e-3
    bound = ops._plain_bound(got)
    assert float(bound) >= float(got.abs().max()) * (1 - 1e-6) and float(bound) <= float(got.abs().max()) * 1.001


@pytest.mark.parametrize("case", [(1, 8, 128, 5, 64, 64, 1, 1), (2, 2, 64, 3, 21, 40, 0, 0), (1, 4, 36, 5, 33, 47, 1, 2), (1, 8, 128, 5, 256, 256, 1, 1)],
                         ids=lambda c: f"B{c[0]}_{c[1]}to{c[2]}_k{c[3]}_{c[4]}x{c[5]}")
def test_conv_sbs_precision16_is_fp16_operands_with_wide_sums(case, dev):
    """mrx_conv_sbs_p16 (the first layers: k x k on <= 8 channels into <= 128, conv_layers.py:121-123 -- the qRIM's 5x5 8 -> 128) against the CPU float64
    convolution of the fp16-rounded operands; fp32-class outside the context."""
    from mridc_amd import ops
    B, Cin, Cout, k, H, W, pad_mode, act = case
    g = torch.Generator().manual_seed(Cin * Cout + H)
    x, w, b = torch.randn(B, Cin, H, W, generator=g), torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5, torch.randn(Cout, generator=g)

    def ref_of(xx, ww):
        xp = Fn.pad(xx, (k // 2,) * 4, mode="replicate" if pad_mode == ops.PAD_REPLICATE else "constant")
        y = Fn.conv2d(xp, ww, b.double())
        return Fn.relu(y) if act == ops.ACT_RELU else (Fn.leaky_relu(y, 0.1) if act == ops.ACT_LEAKY else y)

    with ops.inference_precision(16):
        got = ops.conv_sbs(x.to(dev), w.to(dev), b.to(dev), pad_mode, act, 0.1)
    got32 = ops.conv_sbs(x.to(dev), w.to(dev), b.to(dev), pad_mode, act, 0.1)
    assert rel_l2(got, ref_of(_r16(x), _r16(w))) <= OP_TOL, rel_l2(got, ref_of(_r16(x), _r16(w)))
    assert rel_l2(got32, ref_of(x.double(), w.double())) <= 2e-6
    assert 1e-5 <= rel_l2(got, got32) <= 2e-3


def test_qcirim_precision16_against_autocast_and_fp32(dev):
    """qCIRIM (base_qcirim_run.yaml:204 `precision: 16`) with the precision from its cfg: one cascade of eight steps (IndRNN, 128 filters) at 4 echoes x 8 coils x
    64 x 64 -- the 3x3 dilation-2 128 -> 128 convolutions on one fp16 term, the 5x5 / 1x1 layers, the signal model and the FFTs as before -- within SURVEY appendix
    C's 3e-2 of the oracle under torch.autocast(float16), no further from the fp32 oracle than that oracle is, and not the fp32-class route."""
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
    model = qCIRIM(dict(cfg, precision=16)).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rnn.ih.weight") or n.endswith("rnn.hh"):
                p.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    E, C, H, W = 4, 8, 64, 64
    TEs = [3.0, 11.5, 20.0, 28.5]
    g = torch.Generator().manual_seed(42)
    maps = [torch.rand(1, H, W, generator=g) * s for s in (60.0, 1.0, 30.0, 0.5)]
    S = torch.randn(1, C, H, W, 2, generator=g) / C ** 0.5
    mask = (torch.rand(1, 1, 1, 1, W, 1, generator=g) < 0.3)
    y = torch.randn(1, E, C, H, W, 2, generator=g) * mask
    with torch.no_grad():
        ref32 = oracle.qrim.qcirim_forward(sd, cfg, maps[0], maps[1], maps[2], maps[3], TEs, y, S, None, mask)
        with oracle.amp.autocast_fp16():
            refa = oracle.qrim.qcirim_forward(sd, cfg, maps[0], maps[1], maps[2], maps[3], TEs, y, S, None, mask)
    model = model.to(dev)
    args = [m_.to(dev) for m_ in maps] + [TEs, y.to(dev), S.to(dev), None, mask.to(dev)]
    with torch.no_grad():
        out = next(model(*args))
        model.precision = 32
        out32 = next(model(*args))
    last = lambda o: torch.stack([o[1 + m_][-1][-1].float().cpu() for m_ in range(4)]).double()  # noqa: E731  (the four maps after the last step)
    got, got32, w32, wa = last(out), last(out32), last(ref32), last(refa)
    assert rel_l2(got32, w32) <= 5e-5
    e_a, e_32, a_32 = rel_l2(got, wa), rel_l2(got, w32), rel_l2(wa, w32)
    assert e_a <= NET_TOL_AUTOCAST, e_a
    assert e_32 <= max(2.0 * a_32, 2e-4), (e_32, a_32)
    assert e_32 >= 1e-6


def test_sensitivity_network_in_precision16(dev):
    """`use_sens_net: true` under `precision: 16`: the reference estimates the maps inside the same autocast (models/base.py:392, :886-932); the runner wraps
    `model.sens_net` in the model's precision.  BaseSensitivityModel (NormUnet(8, 4) per coil on the low-frequency block, RSS-normalised) inside
    `inference_precision(16)` against the oracle with the restated kernel arithmetic, under autocast, and in fp32; and through `runner.ReconstructionRunner.predict`."""
    from mridc_amd import ops, runner
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    cfg = dict(synthetic.E2EVN_BASELINE_CFG, num_cascades=2, use_sens_net=True, sens_chans=8, sens_pools=4, sens_mask_type="2D", sens_normalize=True,
               sens_mask_center=True, precision=16)
    torch.manual_seed(4)
    model = VarNet(cfg).eval()
    sd = {k[len("sens_net."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("sens_net.")}
    d = synthetic.make_slice(C=6, H=96, W=80, slice_idx=2)
    y, mask = d["y"], d["mask"]
    with torch.no_grad():
        ref32 = oracle.models.sens_net_forward(sd, cfg, y, mask)
        with oracle.amp.fp16_kernel_arithmetic():
            refk = oracle.models.sens_net_forward(sd, cfg, y, mask)
        with oracle.amp.autocast_fp16():
            refa = oracle.models.sens_net_forward(sd, cfg, y, mask).float()
    model = model.to(dev)
    with torch.no_grad():
        with ops.inference_precision(16):
            got = model.sens_net(y.to(dev), mask.to(dev))
        got32 = model.sens_net(y.to(dev), mask.to(dev))
    e_k, e_a, e_32, k_32 = rel_l2(got, refk), rel_l2(got, refa), rel_l2(got, ref32), rel_l2(refk, ref32)
    assert rel_l2(got32, ref32) <= 5e-5
    assert e_k <= NET_TOL_KERNEL and e_a <= NET_TOL_AUTOCAST, (e_k, e_a)
    assert 0.5 * k_32 <= e_32 <= 2.0 * k_32 and e_32 >= 1e-5, (e_32, k_32)
    # through the runner: the maps it hands the model are the precision-16 ones
    seen = {}
    keep = model.forward
    model.forward = lambda y_, S_, *a, **k: (seen.setdefault("S", S_), keep(y_, S_, *a, **k))[1]
    try:
        runner.ReconstructionRunner(model).predict(y.to(dev), d["sensitivity_maps"].to(dev), mask.to(dev), None, d["target"].to(dev), kspace=y.to(dev))
    finally:
        model.forward = keep
    assert torch.equal(seen["S"], got)



def test_g22_hip_precision16_against_the_reference_run_under_autocast(dev):
    """G22 (tests/golden/g22_precision16.npz): the REFERENCE's RIMBlock and VarNetBlock / NormUnet run under torch.autocast(float16) in the build container.  The
    HIP precision-16 routes against those vectors directly -- no oracle in between: RIMBlock.precision = 16 (reference-init weights 1e-4, recurrent weights x 5
    2e-3: the bounds of tests/test_gpu_amp16.py against the autocast oracle) and the VarNet block inside inference_precision(16) (3e-2; the U-Net route keeps its
    activations in fp32 where autocast rounds them)."""
    import json
    from tests._util import Golden, T, meta, weights
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.rim.rim_block import RIMBlock
    from mridc_amd.collections.reconstruction.models.unet_base.unet_block import NormUnet
    from mridc_amd.collections.reconstruction.models.varnet.vn_block import VarNetBlock
    z = Golden()("g22_precision16.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        if nm.startswith("rim_"):
            blk = RIMBlock(**cfg)
            blk.load_state_dict(weights(z, f"{nm}/w/"))
            blk = blk.to(dev).eval()
            blk.precision = 16
            assert blk._amp16_route()
            y, S, mask = T(z[f"{nm}/y"]).to(dev), T(z[f"{nm}/S"]).to(dev), T(z[f"{nm}/mask"]).to(dev)
            with torch.no_grad():
                outs, _ = blk(y, y, S, mask, None, None, 1.0, keep_eta=False)
            got, want, want32 = torch.stack(outs), T(z[f"{nm}/outs"]), T(z[f"{nm}/outs_fp32"])
            tol = 2e-3 if nm.endswith("x5") else 1e-4
            assert rel_l2(got, want) <= tol, (nm, rel_l2(got, want))
            assert rel_l2(got, want32) <= 2 * max(rel_l2(want, want32), tol), (nm, rel_l2(got, want32), rel_l2(want, want32))
        else:
            nu = NormUnet(cfg["chans"], cfg["num_pools"], padding_size=cfg["padding_size"], normalize=cfg["normalize"])
            blk = VarNetBlock(nu, fft_centered=cfg["fft_centered"], fft_normalization=cfg["fft_normalization"], spatial_dims=[-2, -1], coil_dim=1, no_dc=cfg["no_dc"])
            blk.load_state_dict(weights(z, f"{nm}/w/"))
            blk = blk.to(dev).eval()
            pred, y, S, mask = (T(z[f"{nm}/{k}"]).to(dev) for k in ("pred", "y", "S", "mask"))
            with torch.no_grad(), ops.inference_precision(16):
                nu_out = blk.model(T(z[f"{nm}/eta_in"]).to(dev))
                out = blk(pred, y, S, mask)
            assert rel_l2(nu_out, T(z[f"{nm}/normunet_out"])) <= NET_TOL_AUTOCAST, (nm, rel_l2(nu_out, T(z[f"{nm}/normunet_out"])))
            assert rel_l2(out, T(z[f"{nm}/out"])) <= NET_TOL_AUTOCAST, (nm, rel_l2(out, T(z[f"{nm}/out"])))
            assert rel_l2(out, T(z[f"{nm}/out_fp32"])) <= 2 * max(rel_l2(T(z[f"{nm}/out"]), T(z[f"{nm}/out_fp32"])), 1e-4)
    # qCIRIM, one cascade of the model-zoo widths, with the precision from its cfg
    from mridc_amd.collections.quantitative.models.qcirim import qCIRIM
    cfg = meta(z, "qcirim/cfg")
    model = qCIRIM(dict(cfg, precision=16))
    model.load_state_dict(weights(z, "qcirim/w/"))
    model = model.to(dev).eval()
    TEs = [float(t) for t in z["qcirim/TEs"]]
    args = [T(z[f"qcirim/{k}"]).to(dev) for k in ("r2i", "s0i", "b0i", "phi_i")] + [TEs, T(z["qcirim/y"]).to(dev), T(z["qcirim/S"]).to(dev), None, T(z["qcirim/mask"]).to(dev)]
    with torch.no_grad():
        out = next(model(*args))
    ref, ref32 = T(z["qcirim/out"]), T(z["qcirim/out_fp32"])                  # [step, B, 4, H, W]
    got = torch.stack([torch.stack([t.float().cpu() for t in out[1 + m_][0]]) for m_ in range(4)], 2)
    assert rel_l2(got, ref) <= NET_TOL_AUTOCAST, rel_l2(got, ref)
    assert rel_l2(got, ref32) <= 2 * max(rel_l2(ref, ref32), 1e-4), (rel_l2(got, ref32), rel_l2(ref, ref32))
    assert rel_l2(got, ref32) >= 1e-6

