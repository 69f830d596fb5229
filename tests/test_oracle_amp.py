"""oracle/amp.py on the CPU: the two restatements of the reference's mixed-precision training arithmetic (base_cirim_train.yaml:180) that check the
HIP bf16 tape -- torch's own autocast and the kernels' operand-rounding arithmetic -- against the fp32 oracle and against their definitions."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from mridc_amd import synthetic


def _state(cfg, seed, boost):
    from mridc_amd.collections.reconstruction.models.cirim import CIRIM
    torch.manual_seed(seed)
    model = CIRIM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if boost != 1.0 and (n_.endswith("rnn.ih.weight") or n_.endswith("rnn.hh")):
                p_.mul_(boost)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def _flat(g):
    return torch.cat([g[k].reshape(-1).double() for k in sorted(g) if not k.endswith("dc_weight")])


def test_operand_rounding_convolution_matches_its_definition():
    """Forward = the fp64 convolution of the bf16-rounded operands; data / weight gradient = the same with the incoming gradient rounded."""
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 5, 12, 11), torch.randn(7, 5, 3, 3) / 4, torch.randn(7)
    r = oracle.amp.bf16_round
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = oracle.amp._conv2d_bf16_operands(xg, wg, bg, padding=2, dilation=2)
    want = F.conv2d(r(x).double(), r(w).double(), b.double(), padding=2, dilation=2)
    assert float((y.double() - want).abs().max()) <= 1e-6 * float(want.abs().max())
    dy = torch.randn_like(y)
    y.backward(dy)
    xr, wr = r(x).double().requires_grad_(True), r(w).double().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=2, dilation=2).backward(r(dy).double())
    assert float((xg.grad.double() - xr.grad).abs().max()) <= 1e-6 * float(xr.grad.abs().max())
    assert float((wg.grad.double() - wr.grad).abs().max()) <= 1e-6 * float(wr.grad.abs().max())
    assert torch.allclose(bg.grad, dy.sum((0, 2, 3)), rtol=1e-6, atol=1e-6)
    assert r(r(x)).equal(r(x)) and float(((r(x) - x) / x).abs().max()) <= 2.0 ** -8      # round to nearest: half an ulp of 8 significant bits


@pytest.mark.parametrize("seed,boost", [(0, 1.0), (5, 3.0)])
def test_mixed_precision_oracles_against_the_fp32_oracle(seed, boost):
    """The fp32 mode IS the oracle; the two half-precision arithmetics sit 1e-3 .. 1e-1 away from it in the whole gradient vector (a bf16
    implementation cannot be held to the fp32 oracle more tightly than the reference's own AMP arithmetic is) and closer to each other."""
    cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
    state = _state(cfg, seed, boost)
    s = synthetic.make_slice(4, 48, 40, slice_idx=7)
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    pred = oracle.models.cirim_forward(p, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])
    loss = oracle.models.cirim_process_loss(s["target"], pred, torch.nn.L1Loss(), 8, 2)
    loss.backward()
    res = {m: oracle.amp.cirim_loss_and_gradients(state, cfg, s, m) for m in ("fp32", "autocast_bf16", "bf16_operands")}
    assert float(res["fp32"][0]) == float(loss.detach())
    for k, g in res["fp32"][1].items():
        assert g.equal(p[k].grad), k
    f32, amp, opr = (_flat(res[m][1]) for m in ("fp32", "autocast_bf16", "bf16_operands"))
    d = lambda a, b_: float((a - b_).norm() / b_.norm())  # noqa: E731
    assert 1e-3 <= d(amp, f32) <= 1e-1, d(amp, f32)
    assert 1e-3 <= d(opr, f32) <= 1e-1, d(opr, f32)
    assert d(opr, amp) <= 1e-1
    for m in ("autocast_bf16", "bf16_operands"):
        assert abs(float(res[m][0]) - float(loss.detach())) <= 2e-2 * abs(float(loss.detach()))
    assert all(g.dtype == torch.float32 for g in res["autocast_bf16"][1].values())


def test_precision16_inference_checkers_against_their_definitions_and_the_fp32_oracle():
    """The two checkers of the precision-16 inference route (csrc/rim_amp16.hip): `autocast_fp16` is torch's own CPU autocast (the reference's `precision: 16`,
    base_cirim_run.yaml:132: convolutions return half tensors, the recurrence and eta stay fp32); `fp16_kernel_arithmetic` restates the kernels -- a convolution is
    the float64 product of fp16-rounded operands with an fp32 result, a recurrent cell's new state is rounded to fp16 once.  Both sit ~1e-5 .. 1e-3 from the fp32
    oracle on the final image and closer to each other than to it on boosted weights; outside the context the oracle is the fp32 oracle again."""
    import contextlib
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 5, 12, 11), torch.randn(7, 5, 3, 3) / 4, torch.randn(7)
    r16 = oracle.amp.fp16_round
    assert r16(r16(x)).equal(r16(x)) and float(((r16(x) - x) / x).abs().max()) <= 2.0 ** -11            # round to nearest: half an ulp of 11 significant bits
    with oracle.amp.fp16_kernel_arithmetic():
        y = oracle.rim._conv2d(x, w, b, padding=2, dilation=2)
        h = oracle.rim.indrnn_cell(x[:, :5], torch.zeros(2, 7, 12, 11), torch.randn(7, 5, 1, 1), None, torch.ones(1, 7, 1, 1), 1, 1)
    want = F.conv2d(r16(x).double(), r16(w).double(), b.double(), padding=2, dilation=2)
    assert y.dtype == torch.float32 and float((y.double() - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert h.equal(r16(h))                                                                                  # the state is fp16-representable
    assert oracle.rim._CONV2D[0] is F.conv2d and oracle.rim._STATE[0](x) is x                               # hooks restored
    with oracle.amp.autocast_fp16():
        assert F.conv2d(x, w, b, padding=1).dtype == torch.float16 and (torch.ones(1, 7, 1, 1) * F.conv2d(x, w, b, padding=1)).dtype == torch.float32
    for boost in (1.0, 5.0):
        cfg = dict(synthetic.CIRIM_BASELINE_CFG, num_cascades=2)
        state = _state(cfg, 0, boost)
        s = synthetic.make_slice(4, 48, 40, slice_idx=7)
        out = {}
        for name, ctx in (("fp32", contextlib.nullcontext), ("autocast_fp16", oracle.amp.autocast_fp16), ("kernel", oracle.amp.fp16_kernel_arithmetic)):
            with ctx(), torch.no_grad():
                out[name] = torch.view_as_real(oracle.models.cirim_forward(state, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"])[-1][-1].to(torch.complex64)).double()
        rel = lambda a, b_: float((a - b_).norm() / b_.norm())  # noqa: E731
        d_ac, d_k, d_ak = rel(out["autocast_fp16"], out["fp32"]), rel(out["kernel"], out["fp32"]), rel(out["kernel"], out["autocast_fp16"])
        assert 1e-7 < d_ac < 3e-3 and 1e-7 < d_k < 3e-3, (boost, d_ac, d_k)
        assert d_ak < 2.0 * max(d_ac, d_k), (boost, d_ak, d_ac, d_k)



def test_precision16_checkers_of_the_unet_models():
    """The same two checkers for the U-Net models (base_vn_run.yaml:98, base_unet_run.yaml:96 `precision: 16`; mrx_unet_conv3x3_p16): inside
    `fp16_kernel_arithmetic` the 3x3 convolutions of unet_block.py:250-259 multiply fp16-rounded operands with wide sums (transposed and 1x1 convolutions, the
    normalisations, FFTs and the data consistency untouched: what the HIP route leaves in fp32), autocast is torch's own; both 1e-5 .. 1e-2 from the fp32
    oracle on a two-cascade VarNet and as close to each other; the hook is restored."""
    import contextlib
    from oracle import unet as ounet
    torch.manual_seed(1)
    x, w0, w1 = torch.randn(2, 3, 12, 10), torch.randn(6, 3, 3, 3) / 5, torch.randn(6, 6, 3, 3) / 7
    r16 = oracle.amp.fp16_round
    with oracle.amp.fp16_kernel_arithmetic():
        got = ounet.conv_block(x, w0, w1)
    mid = ounet._in_lrelu(F.conv2d(r16(x).double(), r16(w0).double(), padding=1).float())
    want = ounet._in_lrelu(F.conv2d(r16(mid).double(), r16(w1).double(), padding=1).float())
    assert got.dtype == torch.float32 and float((got - want).abs().max()) <= 1e-5
    assert ounet._CONV3X3[0] is F.conv2d
    cfg = dict(synthetic.E2EVN_BASELINE_CFG, num_cascades=2)
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    torch.manual_seed(0)
    state = {k: v.detach().clone() for k, v in VarNet(cfg).state_dict().items()}
    s = synthetic.make_slice(4, 48, 40, slice_idx=3)
    out = {}
    for name, ctx in (("fp32", contextlib.nullcontext), ("autocast_fp16", oracle.amp.autocast_fp16), ("kernel", oracle.amp.fp16_kernel_arithmetic)):
        with ctx(), torch.no_grad():
            out[name] = torch.view_as_real(oracle.models.varnet_forward(state, cfg, s["y"], s["sensitivity_maps"], s["mask"], None, s["target"]).to(torch.complex64)).double()
    rel = lambda a, b_: float((a - b_).norm() / b_.norm())  # noqa: E731
    d_ac, d_k, d_ak = rel(out["autocast_fp16"], out["fp32"]), rel(out["kernel"], out["fp32"]), rel(out["kernel"], out["autocast_fp16"])
    assert 1e-5 < d_ac < 1e-2 and 1e-5 < d_k < 1e-2, (d_ac, d_k)
    assert d_ak < 2.0 * max(d_ac, d_k), (d_ak, d_ac, d_k)


def test_g22_oracle_under_autocast_is_the_reference_under_autocast(golden):
    """G22 (tests/golden/generate_golden.py:g22_precision16): the reference's own RIMBlock and VarNetBlock / NormUnet run under `torch.autocast("cpu", float16)` --
    what `trainer.precision: 16` (base_cirim_run.yaml:132, base_vn_run.yaml:98) wraps the forward pass in.  The oracle inside `oracle.amp.autocast_fp16()` -- the
    checker every HIP precision-16 test compares with -- must reproduce those vectors: the same torch ops in the same dtypes, so to the last bit or within a few
    fp16 roundings of the outputs, and clearly apart from the fp32 outputs stored beside them."""
    import json
    from tests._util import T, meta, weights
    z = golden("g22_precision16.npz")
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        p = weights(z, f"{nm}/w/")
        if nm.startswith("rim_"):
            rc = oracle.rim.RIMConfig(**cfg)
            y, S, mask = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
            with torch.no_grad(), oracle.amp.autocast_fp16():
                outs, hx = oracle.rim.rim_block_forward(p, rc, y, y, S, mask, None, None, 1.0, keep_eta=False)
            got, want, want32 = torch.stack([o.float() for o in outs]), T(z[f"{nm}/outs"]), T(z[f"{nm}/outs_fp32"])
            assert rel(got, want) <= 1e-6, (nm, rel(got, want))
            for j, h in enumerate(hx):
                assert rel(h.float(), T(z[f"{nm}/hx{j}"])) <= 1e-6, (nm, j)
            assert rel(want, want32) >= 10 * max(rel(got, want), 1e-7)                 # (the vectors ARE the half-precision ones)
        else:
            pred, y, S, mask = T(z[f"{nm}/pred"]), T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
            with torch.no_grad(), oracle.amp.autocast_fp16():
                nu = oracle.unet.norm_unet_forward(p, T(z[f"{nm}/eta_in"]), cfg["num_pools"], cfg["padding_size"], cfg["normalize"], prefix="model.unet.")
                out = oracle.varnet.varnet_block_forward(p, pred, y, S, mask, cfg["num_pools"], cfg["padding_size"], cfg["normalize"], cfg["fft_centered"],
                                                         cfg["fft_normalization"], [-2, -1], 1, cfg["no_dc"])
            assert rel(nu.float(), T(z[f"{nm}/normunet_out"])) <= 1e-6, (nm, rel(nu.float(), T(z[f"{nm}/normunet_out"])))
            assert rel(out.float(), T(z[f"{nm}/out"])) <= 1e-6, (nm, rel(out.float(), T(z[f"{nm}/out"])))
            assert rel(T(z[f"{nm}/out"]), T(z[f"{nm}/out_fp32"])) >= 1e-5
    # the one-cascade qCIRIM of the model-zoo widths (base_qcirim_run.yaml:204)
    cfg = meta(z, "qcirim/cfg")
    TEs = [float(t) for t in z["qcirim/TEs"]]
    args = [T(z[f"qcirim/{k}"]) for k in ("r2i", "s0i", "b0i", "phi_i")] + [TEs, T(z["qcirim/y"]), T(z["qcirim/S"]), None, T(z["qcirim/mask"])]
    with torch.no_grad(), oracle.amp.autocast_fp16():
        out = oracle.qrim.qcirim_forward(weights(z, "qcirim/w/"), cfg, *args)
    ref, ref32 = T(z["qcirim/out"]), T(z["qcirim/out_fp32"])                  # [step, B, 4, H, W]
    for m in range(4):
        got = torch.stack([t.float() for t in out[1 + m][0]])
        assert rel(got, ref[:, :, m]) <= 1e-6, (m, rel(got, ref[:, :, m]))
    assert rel(ref, ref32) >= 1e-5



def test_g23_oracle_training_gradients_are_the_reference_under_autocast(golden):
    """G23 (tests/golden/generate_golden.py:g23_training_autocast): the reference's RIMBlocks composed as cirim.py:146-165, the loss of cirim.py:199-247 and torch
    autograd THROUGH THE REFERENCE MODULES, under torch.autocast(bfloat16) -- BASELINE config 4's arithmetic on the CPU -- and in fp32.  The oracle's
    `cirim_loss_and_gradients` (what the HIP training tape is compared with, tests/test_gpu_train_bf16.py / bench.py --train) must give the reference's loss and
    every one of its parameter gradients in both arithmetics."""
    from tests._util import T, meta, weights
    z = golden("g23_training_autocast.npz")
    cfg = meta(z, "cfg")
    state = weights(z, "w/")
    sample = dict(y=T(z["y"]), sensitivity_maps=T(z["S"]), mask=T(z["mask"]), target=T(z["target"]))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    # (measured: fp32 <= 2e-7; under autocast nine of the eleven tensors per cascade to the last bit, the two 64-channel convolution weights 3e-5 / 6e-5 -- the
    # order of one bf16 sum -- while the two arithmetics themselves are 4e-2 apart)
    for mode, tag, tol in (("fp32", "fp32", 2e-6), ("autocast_bf16", "autocast_bf16", 2e-4)):
        loss, grads = oracle.amp.cirim_loss_and_gradients(state, cfg, sample, mode=mode)
        want_loss = float(T(z[f"loss_{tag}"])[0])
        assert abs(float(loss) - want_loss) <= 1e-6 * abs(want_loss), (mode, float(loss), want_loss)
        names = [k[len(f"grad_{tag}/"):] for k in z.files if k.startswith(f"grad_{tag}/")]
        assert sorted(names) == sorted(grads), (mode, sorted(set(names) ^ set(grads)))
        for n in names:
            assert rel(grads[n], T(z[f"grad_{tag}/{n}"])) <= tol, (mode, n, rel(grads[n], T(z[f"grad_{tag}/{n}"])))
    g16 = torch.cat([T(z[k]).reshape(-1) for k in sorted(z.files) if k.startswith("grad_autocast_bf16/")])
    g32 = torch.cat([T(z[k]).reshape(-1) for k in sorted(z.files) if k.startswith("grad_fp32/")])
    assert rel(g16, g32) >= 1e-3                                                      # (the two arithmetics ARE apart: the vectors are not the fp32 ones twice)
