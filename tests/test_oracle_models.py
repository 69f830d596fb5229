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
