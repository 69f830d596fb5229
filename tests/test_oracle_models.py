"""CPU: the oracle's block/model restatements against reference-generated goldens (G5-G8)."""
import json

import torch

import oracle
from tests._util import T, assert_close, meta, weights


def test_g5_rimblock(golden):
    z = golden("g5_rimblock.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        m = meta(z, f"{nm}/meta")
        rc = oracle.rim.RIMConfig(**cfg)
        p = weights(z, f"{nm}/w/")
        y, S, mask = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
        if m["keep_eta"]:
            p0 = T(z[f"{nm}/pred"])
            pred = [p0 * 0.5, p0] if m["pred_is_list"] else p0
        else:
            pred = y
        outs, hx = oracle.rim.rim_block_forward(p, rc, pred, y, S, mask, None, None, 1.0, keep_eta=m["keep_eta"])
        assert_close(torch.stack(outs), T(z[f"{nm}/outs"]), 2e-6, f"{nm} outs")
        for j, h in enumerate(hx):
            assert_close(h, T(z[f"{nm}/hx{j}"]), 2e-6, f"{nm} hx{j}")


def test_g6_cirim(golden):
    z = golden("g6_cirim.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        p = weights(z, f"{nm}/w/")
        y, S, mask, target = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"]), T(z[f"{nm}/target"])
        out = oracle.models.cirim_forward(p, cfg, y, S, mask, None, target)
        ref = T(z[f"{nm}/out"])
        assert len(out) == cfg["num_cascades"] and len(out[0]) == oracle.models.cirim_time_steps(cfg["time_steps"])
        got = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
        assert_close(got, ref, 5e-6, f"{nm} cirim")
        T_ = oracle.models.cirim_time_steps(cfg["time_steps"])
        loss = oracle.models.cirim_process_loss(target, out, torch.nn.L1Loss(), T_, cfg["num_cascades"])
        assert abs(float(loss) - float(z[f"{nm}/loss_l1"][0])) <= 1e-5 * abs(float(z[f"{nm}/loss_l1"][0]))


def test_g7_varnet(golden):
    z = golden("g7_varnet.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        p = weights(z, f"{nm}/w/")
        pred, y, S, mask = T(z[f"{nm}/pred"]), T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
        eta_in = oracle.varnet.sens_reduce(pred, S, cfg["fft_centered"], cfg["fft_normalization"], [-2, -1], 1)
        assert_close(eta_in, T(z[f"{nm}/eta_in"]), 1e-6, f"{nm} sens_reduce")
        nu = oracle.unet.norm_unet_forward(p, T(z[f"{nm}/eta_in"]), cfg["num_pools"], cfg["padding_size"],
                                           cfg["normalize"], prefix="model.unet.")
        assert_close(nu, T(z[f"{nm}/normunet_out"]), 5e-6, f"{nm} normunet")
        out = oracle.varnet.varnet_block_forward(p, pred, y, S, mask, cfg["num_pools"], cfg["padding_size"],
                                                 cfg["normalize"], cfg["fft_centered"], cfg["fft_normalization"],
                                                 [-2, -1], 1, cfg["no_dc"])
        assert_close(out, T(z[f"{nm}/out"]), 5e-6, f"{nm} block")


def test_g8_models(golden):
    z = golden("g8_models.npz")
    cfg = meta(z, "vn/cfg")
    y, S, mask, target = T(z["vn/y"]), T(z["vn/S"]), T(z["vn/mask"]), T(z["vn/target"])
    out = oracle.models.varnet_forward(weights(z, "vn/w/"), cfg, y, S, mask, None, target)
    assert_close(torch.view_as_real(out), T(z["vn/out"]), 1e-5, "varnet")
    ucfg = meta(z, "unet/cfg")
    out = oracle.models.unet_model_forward(weights(z, "unet/w/"), ucfg, T(z["unet/y"]), S, mask, None, target)
    assert_close(torch.view_as_real(out), T(z["unet/out"]), 1e-5, "unet")
    for meth in ("SENSE", "RSS"):
        c = dict(ucfg, coil_combination_method=meth)
        out = oracle.models.zf_forward(c, T(z["unet/y"]), S, mask, target)
        out = torch.view_as_real(out) if out.is_complex() else out
        assert_close(out, T(z[f"zf/out_{meth}"]), 1e-6, f"zf {meth}")


def test_g9_qrim(golden):
    """qMRI: signal model, analytical gradient, qRIMBlock and the 2-cascade qCIRIM composition (A19)."""
    z = golden("g9_qrim.npz")
    TEs = [float(t) for t in z["TEs"]]
    r2, s0, b0, ph = T(z["r2"]), T(z["s0"]), T(z["b0"]), T(z["ph"])
    assert_close(oracle.qrim.megre_signal(r2, s0, b0, ph, TEs), T(z["signal"]), 1e-6, "MEGRE signal")
    S, mask = T(z["S"]), T(z["mask"])
    r2i, s0i, b0i, phi_i = T(z["r2i"]), T(z["s0i"]), T(z["b0i"]), T(z["phi_i"])
    for cen, norm in ((True, "ortho"), (False, "backward")):
        yy = T(z[f"grad/{int(cen)}_{norm}/y"])
        got = torch.stack([oracle.qrim.analytical_log_likelihood_gradient(r2i[i], s0i[i], b0i[i], phi_i[i], TEs, S[i], yy[i],
                                                                          mask[i], cen, norm, [-2, -1], 2) for i in range(2)])
        assert_close(got, T(z[f"grad/{int(cen)}_{norm}/out"]), 2e-6, f"analytical gradient {cen} {norm}")
    cfg = meta(z, "qcirim/cfg")
    out = oracle.qrim.qcirim_forward(weights(z, "qcirim/w/"), cfg, r2i, s0i, b0i, phi_i, TEs, T(z["y"]), S, None, mask)
    ref = T(z["qcirim/out"])                                   # [cascade, step, B, 4, H, W]
    for m in range(4):
        got = torch.stack([torch.stack(c) for c in out[1 + m]])
        assert_close(got, ref[:, :, :, m], 1e-5, f"qcirim map {m}")


def test_g19_cirim_spec_and_harness_metrics(golden):
    """The fixture SURVEY 8c specifies for G6 (model-zoo CIRIM, 8 cascades x 8 steps, 64 filters, [1,15,64,48,2]) and the row-H harness
    outputs: `abs / max` image and MSE / NMSE / SSIM / PSNR with maxval = output.max() - output.min() (models/base.py:415-436)."""
    z = golden("g19_cirim_spec.npz")
    cfg = meta(z, "cfg")
    p = weights(z, "w/")
    y, S, mask, target = T(z["y"]), T(z["S"]), T(z["mask"]), T(z["target"])
    out = oracle.models.cirim_forward(p, cfg, y, S, mask, None, target)
    assert len(out) == 8 and len(out[0]) == 8
    got = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
    assert_close(got, T(z["out"]), 5e-6, "g19 cirim chain")
    o, t = oracle.metrics.postprocess(out[-1][-1], target)
    assert_close(o, T(z["harness/output"]), 1e-6, "harness output")
    assert_close(t, T(z["harness/target"]), 1e-6, "harness target")
    m = oracle.metrics.slice_metrics(out[-1][-1], target)
    want = z["harness/metrics"]                                     # MSE, NMSE, SSIM, PSNR, maxval
    for k, w in zip(("mse", "nmse", "ssim", "psnr"), want[:4]):
        assert abs(m[k] - float(w)) <= 1e-5 * max(1.0, abs(float(w))), (k, m[k], float(w))


def test_g20_rimblock_3d(golden):
    """A14: the 3-D mode of RIMBlock (Conv3d layers over the folded slices) against the reference-generated fixture."""
    z = golden("g20_rim3d.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        rc = oracle.rim.RIMConfig(**cfg)
        p = weights(z, f"{nm}/w/")
        y, S, mask = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
        outs, hx = oracle.rim.rim_block_forward(p, rc, y, y, S, mask, None, None, 1.0, keep_eta=False)
        assert_close(torch.stack(outs), T(z[f"{nm}/outs"]), 2e-6, f"{nm} outs")
        for j, h in enumerate(hx):
            assert_close(h, T(z[f"{nm}/hx{j}"]), 2e-6, f"{nm} hx{j}")
