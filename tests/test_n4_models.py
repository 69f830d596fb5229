"""the other cascades / data-consistency formulations that reuse the hot-path kernels -- CascadeNet
(conv/conv2d.py, cascadenet/ccnn_block.py, ccnn.py), VSNet (variablesplittingnet/vsnet_block.py, vsnet.py) and the sigmanet
data-consistency layers (sigmanet/dc_layers.py).  CPU: the oracle restatement against the reference-generated goldens
G15-G17.  GPU: the HIP-backed drop-ins against the same goldens."""
import json

import pytest
import torch

import oracle
from tests._util import T, assert_close, meta, weights


# ---- CPU: oracle vs goldens -------------------------------------------------------------------------------------------
def test_oracle_cascadenet_vs_golden(golden):
    z = golden("g15_cascadenet.npz")
    for nm in json.loads(str(z["names"])):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        conv_p = {k[len("model."):]: v for k, v in p.items() if k.startswith("model.")}
        out = oracle.cascadenet.conv2d_stack_forward(conv_p, T(z[f"{nm}/conv_in"]), cfg["n_convs"], cfg["batchnorm"])
        assert_close(out, T(z[f"{nm}/conv_out"]), 1e-6, f"{nm} conv stack")
        x5 = T(z[f"{nm}/conv_in"]).permute(0, 2, 3, 1).unsqueeze(1).contiguous()
        assert_close(oracle.cascadenet.conv2d_stack_forward(conv_p, x5, cfg["n_convs"], cfg["batchnorm"]), T(z[f"{nm}/conv_out_5d"]),
                     1e-6, f"{nm} conv stack, 5-D input")
        got = oracle.cascadenet.cascadenet_block_forward(p, T(z[f"{nm}/pred"]), T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"]),
                                                         cfg["n_convs"], cfg["batchnorm"], cfg["fft_centered"],
                                                         cfg["fft_normalization"], [-2, -1], 1, cfg["no_dc"])
        assert_close(got, T(z[f"{nm}/out"]), 2e-6, f"{nm} block")
    cfg = meta(z, "model/cfg")
    got = oracle.cascadenet.cascadenet_forward(weights(z, "model/w/"), cfg, T(z["model/y"]), T(z["model/S"]), T(z["model/mask"]), None,
                                               T(z["model/target"]))
    assert_close(got, T(z["model/out"]), 5e-6, "CascadeNet model")


def test_oracle_vsnet_vs_golden(golden):
    z = golden("g16_vsnet.npz")
    for nm in json.loads(str(z["names"])):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        y, S, mask = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
        got = oracle.vsnet.vsnet_block_forward(p, y, S, mask, cfg["num_cascades"], cfg["imspace_conv_n_convs"], cfg["fft_centered"],
                                               cfg["fft_normalization"], [-2, -1], 1, prefix="model.")
        assert_close(got, T(z[f"{nm}/block_out"]), 5e-6, f"{nm} VSNet block")
        assert_close(oracle.vsnet.data_consistency(got, y, mask, p["model.data_consistency_block.0.dc_weight"]), T(z[f"{nm}/dc"]), 5e-6,
                     f"{nm} hard DC")
        assert_close(oracle.vsnet.weighted_average(got, y, p["model.weighted_average_block.0.param"]), T(z[f"{nm}/wa"]), 5e-6,
                     f"{nm} weighted average")
        out = oracle.vsnet.vsnet_forward(p, cfg, y, S, mask, None, T(z[f"{nm}/target"]))
        assert_close(out, T(z[f"{nm}/model_out"]), 5e-6, f"{nm} VSNet model")


def test_oracle_dc_layers_vs_golden(golden):
    z = golden("g17_dc_layers.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        c, n = cfg["fft_centered"], cfg["fft_normalization"]
        x, y, S, mask = (T(z[f"{nm}/{k}"]) for k in ("x", "y", "S", "mask"))
        assert_close(oracle.dc_layers.data_gd(x, y, S, mask, 0.3, c, n, [-2, -1]), T(z[f"{nm}/gd"]), 2e-6, f"{nm} GD")
        assert_close(oracle.dc_layers.data_vs(x, y, S, mask, 0.4, 0.7, c, n, [-2, -1]), T(z[f"{nm}/vs"]), 2e-6, f"{nm} VS")
        assert_close(oracle.dc_layers.dc_single(x, y[:, 0], mask[:, 0], 0.2, c, n, [-2, -1]), T(z[f"{nm}/dc_single"]), 2e-6, f"{nm} DCLayer")
        for it in (3, 10):
            got = oracle.dc_layers.data_prox_cg(x.unsqueeze(1), y, S, mask, 0.5, 1e-6, it, c, n, [-2, -1])
            assert_close(got, T(z[f"{nm}/prox{it}"]), 2e-5, f"{nm} prox-CG {it} iterations")


def test_oracle_dunet_vs_golden(golden):
    z = golden("g21_dunet.npz")
    for nm in ("didn_a", "didn_b"):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        got = oracle.dunet.didn_forward(p, T(z[f"{nm}/x"]), cfg["num_dubs"], cfg["num_convs_recon"], skip_connection=cfg["skip_connection"])
        assert_close(got, T(z[f"{nm}/out"]), 2e-6, f"{nm} DIDN")
    for nm in json.loads(str(z["names"])):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        y, S, mask = T(z[f"{nm}/y"]), T(z[f"{nm}/S"]), T(z[f"{nm}/mask"])
        reg0 = oracle.dunet.complex_norm_wrapper(
            lambda t: oracle.dunet.didn_forward(p, t, cfg["didn_num_dubs"], cfg["didn_num_convs_recon"], prefix="model.gradR.0.model."),
            T(z[f"{nm}/init_pred"]))
        assert_close(reg0, T(z[f"{nm}/reg0"]), 5e-6, f"{nm} normalisation wrapper, 4-D iterate")
        net = oracle.dunet.sensitivity_network_forward(p, cfg, T(z[f"{nm}/init_pred"]), y, S, mask)
        assert_close(net, T(z[f"{nm}/net_out"]), 2e-5, f"{nm} SensitivityNetwork")
        out = oracle.dunet.dunet_forward(p, cfg, y, S, mask, None, T(z[f"{nm}/target"]))
        assert_close(out, T(z[f"{nm}/model_out"]), 2e-5, f"{nm} DUNet")


def test_oracle_rvn_vs_golden(golden):
    z = golden("g18_rvn.npz")
    for nm in ("gru_h16_l2", "gru_h8_l4", "gru_h8_l2_k3"):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        x, st = T(z[f"{nm}/x"]), T(z[f"{nm}/state"])
        o0, s0 = oracle.rvn.conv2dgru_forward(p, x, None, cfg["num_layers"], cfg["hidden_channels"])
        o1, s1 = oracle.rvn.conv2dgru_forward(p, x, st, cfg["num_layers"], cfg["hidden_channels"])
        for got, key in ((o0, "out0"), (s0, "state0"), (o1, "out1"), (s1, "state1")):
            assert_close(got, T(z[f"{nm}/{key}"]), 2e-6, f"{nm} {key}")
    for nm in ("init_ms1", "init_ms3"):
        cfg = meta(z, f"{nm}/cfg")
        got = oracle.rvn.recurrent_init_forward(weights(z, f"{nm}/w/"), T(z[f"{nm}/x"]), cfg["dilations"], cfg["depth"],
                                                cfg["multiscale_depth"])
        assert_close(got, T(z[f"{nm}/out"]), 2e-6, nm)
    for nm in ("model_shared", "model_unshared"):
        cfg, p = meta(z, f"{nm}/cfg"), weights(z, f"{nm}/w/")
        y, S, mask, target = (T(z[f"{nm}/{k}"]) for k in ("y", "S", "mask", "target"))
        k0, st0 = oracle.rvn.rvn_block_forward(p, y, y, mask, S, T(z[f"{nm}/init_state"]), cfg["recurrent_num_layers"],
                                               cfg["recurrent_hidden_channels"], cfg["fft_centered"], cfg["fft_normalization"],
                                               [-2, -1], 1, prefix="block_list.0.")
        assert_close(k0, T(z[f"{nm}/k_step0"]), 5e-6, f"{nm} block, step 0")
        assert_close(st0, T(z[f"{nm}/state_step0"]), 5e-6, f"{nm} state, step 0")
        assert_close(oracle.rvn.rvn_forward(p, cfg, y, S, mask, None, target), T(z[f"{nm}/out"]), 2e-5, f"{nm} RecurrentVarNet")


# ---- GPU: the HIP-backed drop-ins vs the same goldens ----------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_cascadenet_vs_golden(golden, dev):
    from mridc_amd.collections.reconstruction.models.cascadenet.ccnn_block import CascadeNetBlock
    from mridc_amd.collections.reconstruction.models.ccnn import CascadeNet
    from mridc_amd.collections.reconstruction.models.conv.conv2d import Conv2d
    z = golden("g15_cascadenet.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        net = Conv2d(2, 2, cfg["hidden_channels"], n_convs=cfg["n_convs"], activation=torch.nn.PReLU(), batchnorm=cfg["batchnorm"])
        blk = CascadeNetBlock(net, fft_centered=cfg["fft_centered"], fft_normalization=cfg["fft_normalization"],
                              spatial_dims=[-2, -1], coil_dim=1, no_dc=cfg["no_dc"])
        blk.load_state_dict(weights(z, f"{nm}/w/"))
        blk = blk.to(dev).eval()
        with torch.no_grad():
            x = T(z[f"{nm}/conv_in"]).to(dev)
            assert_close(blk.model(x), T(z[f"{nm}/conv_out"]), 1e-5, f"{nm} conv stack")
            assert_close(blk.model(x.permute(0, 2, 3, 1).unsqueeze(1).contiguous()), T(z[f"{nm}/conv_out_5d"]), 1e-5,
                         f"{nm} conv stack, 5-D input")
            out = blk(T(z[f"{nm}/pred"]).to(dev), T(z[f"{nm}/y"]).to(dev), T(z[f"{nm}/S"]).to(dev), T(z[f"{nm}/mask"]).to(dev))
        assert_close(out, T(z[f"{nm}/out"]), 2e-5, f"{nm} CascadeNetBlock")
        if cfg["batchnorm"]:
            blk.train()
            with pytest.raises(NotImplementedError):
                blk.model(x)
    cfg = meta(z, "model/cfg")
    model = CascadeNet(cfg)
    missing, unexpected = model.load_state_dict(weights(z, "model/w/"), strict=False)
    assert unexpected == [] and missing == ["dc_weight"]
    model = model.to(dev).eval()
    with torch.no_grad():
        out = model(T(z["model/y"]).to(dev), T(z["model/S"]).to(dev), T(z["model/mask"]).to(dev), None, T(z["model/target"]).to(dev))
    assert_close(out, T(z["model/out"]), 5e-5, "CascadeNet model")


@pytest.mark.gpu
def test_vsnet_vs_golden(golden, dev):
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.vsnet import VSNet
    z = golden("g16_vsnet.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        model = VSNet(cfg)
        sd_ = weights(z, f"{nm}/w/")
        assert set(sd_) == set(model.state_dict()), "state_dict layout (shared modules repeat under every cascade index)"
        model.load_state_dict(sd_)
        model = model.to(dev).eval()
        y, S, mask, target = (T(z[f"{nm}/{k}"]).to(dev) for k in ("y", "S", "mask", "target"))
        with torch.no_grad():
            blk_out = model.model(y, S, mask)
            out = model(y, S, mask, None, target)
            dc = model.model.data_consistency_block[0](blk_out, y, mask)
            wa = model.model.weighted_average_block[0](blk_out, y)
        assert_close(blk_out, T(z[f"{nm}/block_out"]), 5e-5, f"{nm} VSNetBlock")
        assert_close(out, T(z[f"{nm}/model_out"]), 5e-5, f"{nm} VSNet")
        assert_close(dc, T(z[f"{nm}/dc"]), 5e-5, f"{nm} hard DC")
        assert_close(wa, T(z[f"{nm}/wa"]), 5e-5, f"{nm} weighted average")
        # the two pointwise kernels round exactly like the reference's separate torch ops
        w = model.model.data_consistency_block[0].dc_weight
        ref_dc = ((1 - mask) * blk_out + mask * y) * w
        assert torch.equal(dc, ref_dc), "hard DC is bit-exact vs the torch expression"
        p = model.model.weighted_average_block[0].param
        sx = model.model.sens_reduce(dc, S)
        ref_wa = p * (y + blk_out) + (1 - p) * sx
        assert torch.equal(ops.vs_average(y, blk_out, sx, p), ref_wa), "weighted average is bit-exact vs the torch expression"
        with pytest.raises(RuntimeError):
            ops.hard_dc(blk_out, y, mask.bool(), w)


@pytest.mark.gpu
def test_sigmanet_dc_layers_vs_golden(golden, dev):
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.sigmanet import dc_layers
    z = golden("g17_dc_layers.npz")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        kw = dict(fft_centered=cfg["fft_centered"], fft_normalization=cfg["fft_normalization"], spatial_dims=[-2, -1])
        x, y, S, mask = (T(z[f"{nm}/{k}"]).to(dev) for k in ("x", "y", "S", "mask"))
        with torch.no_grad():
            gd = dc_layers.DataGDLayer(0.3, **kw).to(dev)(x, y, S, mask)
            vs = dc_layers.DataVSLayer(0.4, 0.7, **kw).to(dev)(x, y, S, mask)
            dc = dc_layers.DCLayer(0.2, **kw).to(dev)(x, y[:, 0], mask[:, 0])
        assert_close(gd, T(z[f"{nm}/gd"]), 1e-5, f"{nm} DataGDLayer")
        assert_close(vs, T(z[f"{nm}/vs"]), 1e-5, f"{nm} DataVSLayer")
        assert_close(dc, T(z[f"{nm}/dc_single"]), 1e-5, f"{nm} DCLayer")
        # pointwise kernels vs the torch expressions they replace
        k = torch.randn(1, 3, 9, 7, 2, device=dev)
        m = (torch.rand(1, 1, 9, 7, 1, device=dev) < 0.5).float()
        assert_close(ops.coil_sum(k, m), (k * m).sum(1), 1e-6, "coil_sum")
        a = torch.randn(1, 9, 7, 2, device=dev)
        assert torch.equal(ops.dc_bcast(a, k, m), (a.unsqueeze(1) - k) * m)
        al = torch.tensor([0.3], device=dev)
        assert torch.equal(ops.dc_bcast(a, k, m, al), (1 - m) * a.unsqueeze(1) + m * (al * a.unsqueeze(1) + (1 - al) * k))
        g = torch.randn(3, 9, 7, 2, device=dev)
        assert torch.equal(ops.lincomb(a, g, al, 0), a - al * g)
        assert torch.equal(ops.lincomb(a, g, al, 1), al * a + (1 - al) * g)
        with pytest.raises(NotImplementedError):
            dc_layers.DataGDLayer(0.3, **kw).to(dev)(x.repeat(2, 1, 1, 1), y, S, mask)
        for it in (3, 10):
            with torch.no_grad():
                px = dc_layers.DataProxCGLayer(0.5, tol=1e-6, iter=it, **kw).to(dev)(x.unsqueeze(1), y, S, mask)
            assert_close(px, T(z[f"{nm}/prox{it}"]), 1e-4, f"{nm} DataProxCGLayer, {it} iterations")
        with pytest.raises(NotImplementedError):
            dc_layers.DataProxCGLayer(0.5, **kw).to(dev)(x, y, S, mask)
        a5, b5 = torch.randn(2, 3, 9, 7, 2, device=dev), torch.randn(2, 3, 9, 7, 2, device=dev)
        ca, cb = torch.view_as_complex(a5), torch.view_as_complex(b5)
        assert_close(ops.cdot(a5, b5), torch.view_as_real((ca * cb.conj()).reshape(2, -1).sum(-1)), 1e-5, "cdot")
    assert list(dc_layers.DataGDLayer(0.3).state_dict()) == ["data_weight"]
    assert list(dc_layers.DataVSLayer(0.1, 0.2).state_dict()) == ["alpha", "beta"]
    assert list(dc_layers.DCLayer(0.1).state_dict()) == ["lambda_"]
    assert list(dc_layers.DataProxCGLayer(0.1).state_dict()) == ["lambdaa"]


@pytest.mark.gpu
def test_dunet_vs_golden(golden, dev):
    """N4: DIDN, the complex-instance-norm wrapper, SensitivityNetwork and the DUNet forward against the reference-generated G21."""
    from mridc_amd.collections.reconstruction.models.didn.didn import DIDN
    from mridc_amd.collections.reconstruction.models.dunet import DUNet
    z = golden("g21_dunet.npz")
    for nm in ("didn_a", "didn_b"):
        cfg = meta(z, f"{nm}/cfg")
        net = DIDN(2, 2, **cfg)
        sd_ = weights(z, f"{nm}/w/")
        assert set(sd_) == set(net.state_dict()), "DIDN state_dict layout (the doubled conv / PReLU pairs included)"
        net.load_state_dict(sd_)
        net = net.to(dev).eval()
        with torch.no_grad():
            out = net(T(z[f"{nm}/x"]).to(dev))
        assert_close(out, T(z[f"{nm}/out"]), 2e-5, f"{nm} DIDN")
    for nm in json.loads(str(z["names"])):
        cfg = meta(z, f"{nm}/cfg")
        model = DUNet(cfg)
        sd_ = weights(z, f"{nm}/w/")
        sd_["dc_weight"] = torch.ones(1)
        assert set(sd_) == set(model.state_dict())
        model.load_state_dict(sd_)
        model = model.to(dev).eval()
        y, S, mask, target = (T(z[f"{nm}/{k}"]).to(dev) for k in ("y", "S", "mask", "target"))
        with torch.no_grad():
            reg0 = model.model.gradR[0](T(z[f"{nm}/init_pred"]).to(dev))
            net_out = model.model(T(z[f"{nm}/init_pred"]).to(dev), y, S, mask)
            out = model(y, S, mask, None, target)
        assert_close(reg0, T(z[f"{nm}/reg0"]), 5e-5, f"{nm} normalisation wrapper")
        assert_close(net_out, T(z[f"{nm}/net_out"]), 2e-4, f"{nm} SensitivityNetwork")
        assert_close(out, T(z[f"{nm}/model_out"]), 2e-4, f"{nm} DUNet")
    with pytest.raises(NotImplementedError):
        DUNet(dict(meta(z, "prox_shared/cfg"), reg_model_architecture="MWCNN"))
    with pytest.raises(ValueError):
        DUNet(dict(meta(z, "prox_shared/cfg"), train_loss_fn="huber"))


@pytest.mark.gpu
def test_rvn_vs_golden(golden, dev):
    from mridc_amd.collections.reconstruction.models.recurrentvarnet.conv2gru import Conv2dGRU
    from mridc_amd.collections.reconstruction.models.recurrentvarnet.recurrentvarnet import RecurrentInit
    from mridc_amd.collections.reconstruction.models.rvn import RecurrentVarNet
    z = golden("g18_rvn.npz")
    for nm in ("gru_h16_l2", "gru_h8_l4", "gru_h8_l2_k3"):
        cfg = meta(z, f"{nm}/cfg")
        net = Conv2dGRU(cfg["in_channels"], cfg["hidden_channels"], num_layers=cfg["num_layers"], gru_kernel_size=cfg["gru_kernel_size"],
                        replication_padding=True)
        net.load_state_dict(weights(z, f"{nm}/w/"))
        net = net.to(dev).eval()
        x, st = T(z[f"{nm}/x"]).to(dev), T(z[f"{nm}/state"]).to(dev)
        with torch.no_grad():
            o0, s0 = net(x, None)
            o1, s1 = net(x, st)
            o2, s2 = net(x, [st[..., i].contiguous() for i in range(cfg["num_layers"])])
        for got, key in ((o0, "out0"), (s0, "state0"), (o1, "out1"), (s1, "state1"), (o2, "out1"), (torch.stack(s2, -1), "state1")):
            assert_close(got, T(z[f"{nm}/{key}"]), 1e-5, f"{nm} {key}")
    for nm in ("init_ms1", "init_ms3"):
        cfg = meta(z, f"{nm}/cfg")
        ini = RecurrentInit(cfg["in_channels"], cfg["out_channels"], channels=tuple(cfg["channels"]), dilations=tuple(cfg["dilations"]),
                            depth=cfg["depth"], multiscale_depth=cfg["multiscale_depth"])
        ini.load_state_dict(weights(z, f"{nm}/w/"))
        with torch.no_grad():
            assert_close(ini.to(dev)(T(z[f"{nm}/x"]).to(dev)), T(z[f"{nm}/out"]), 1e-5, nm)
    for nm in ("model_shared", "model_unshared"):
        cfg = meta(z, f"{nm}/cfg")
        model = RecurrentVarNet(cfg)
        model.load_state_dict(weights(z, f"{nm}/w/"))
        model = model.to(dev).eval()
        y, S, mask, target = (T(z[f"{nm}/{k}"]).to(dev) for k in ("y", "S", "mask", "target"))
        with torch.no_grad():
            k0, st0 = model.block_list[0](y, y, mask, S, T(z[f"{nm}/init_state"]).to(dev))
            out = model(y, S, mask, None, target)
        assert_close(k0, T(z[f"{nm}/k_step0"]), 2e-5, f"{nm} block, step 0")
        assert_close(st0, T(z[f"{nm}/state_step0"]), 2e-5, f"{nm} state, step 0")
        assert_close(out, T(z[f"{nm}/out"]), 1e-4, f"{nm} RecurrentVarNet")


@pytest.mark.gpu
def test_conv2dgru_64_one_launch_cell(dev, monkeypatch):
    """The Recurrent VarNet's own shape (1x1 gates, 64 features): the one-launch GRU against the oracle and against the unfused
    route of the same module; pixel count not a multiple of the 32-pixel wave segment; zero state as None."""
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.recurrentvarnet.conv2gru import Conv2dGRU
    torch.manual_seed(31)
    net = Conv2dGRU(2, 64, num_layers=3, replication_padding=True).eval()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0, 0.2)
    p = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 2, 21, 19, generator=g)
    st = torch.randn(2, 64, 21, 19, 3, generator=g) * 0.5
    ref0 = oracle.rvn.conv2dgru_forward(p, x, None, 3, 64)
    ref1 = oracle.rvn.conv2dgru_forward(p, x, st, 3, 64)
    net = net.to(dev)
    assert ops.conv2dgru_supported(64, 64, 1) and not ops.conv2dgru_supported(16, 16, 1) and not ops.conv2dgru_supported(64, 64, 3)
    with torch.no_grad():
        got0, got1 = net(x.to(dev), None), net(x.to(dev), st.to(dev))
        monkeypatch.setattr(ops, "conv2dgru_supported", lambda *a: False)
        unf1 = net(x.to(dev), st.to(dev))
    for (go, gs), (ro, rs), what in ((got0, ref0, "zero state"), (got1, ref1, "given state")):
        assert_close(go, ro, 2e-5, f"Conv2dGRU-64 output, {what}")
        assert_close(gs, rs, 2e-5, f"Conv2dGRU-64 states, {what}")
    assert_close(got1[0], unf1[0], 2e-5, "one launch vs unfused, output")
    assert_close(got1[1], unf1[1], 2e-5, "one launch vs unfused, states")


@pytest.mark.gpu
def test_hybrid_space_cascades_equal_kspace_cascades(golden, dev, monkeypatch):
    """Row-invariant masks: VarNet / CascadeNet run their cascades on IFFT_H(k) with row transforms only.  Same outputs as the k-space
    form (MRIDC_AMD_HYBRID=0), which is the one the goldens pin; a 2-D mask takes the k-space form by itself."""
    from mridc_amd import ops
    from mridc_amd.collections.reconstruction.models.ccnn import CascadeNet
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    z8, z15 = golden("g8_models.npz"), golden("g15_cascadenet.npz")
    for cls, z, pre in ((VarNet, z8, "vn"), (CascadeNet, z15, "model")):
        cfg = meta(z, f"{pre}/cfg")
        model = cls(cfg)
        model.load_state_dict(weights(z, f"{pre}/w/"), strict=False)
        model = model.to(dev).eval()
        y, S, mask, target = (T(z[f"{pre}/{k}"]).to(dev) for k in ("y", "S", "mask", "target"))
        assert model._hybrid_ok(mask)
        with torch.no_grad():
            hyb = model(y, S, mask, None, target)
            monkeypatch.setenv("MRIDC_AMD_HYBRID", "0")
            assert not model._hybrid_ok(mask)
            ksp = model(y, S, mask, None, target)
            monkeypatch.delenv("MRIDC_AMD_HYBRID")
        assert_close(hyb, ksp, 2e-5, f"{cls.__name__}: hybrid-space vs k-space cascades")
        assert_close(hyb, T(z[f"{pre}/out"]), 1e-4, f"{cls.__name__}: hybrid-space cascades vs the reference")
        m2 = (torch.rand(1, 1, y.shape[2], y.shape[3], 1, device=dev) < 0.4)
        assert not model._hybrid_ok(m2) and not ops.mask_is_row_invariant(m2)
    # the Recurrent VarNet takes the same route (VSNet cannot: it adds an image to k-space, vsnet_block.py:145)
    from mridc_amd.collections.reconstruction.models.rvn import RecurrentVarNet
    for cls, z, pre in ((RecurrentVarNet, golden("g18_rvn.npz"), "model_shared"), (RecurrentVarNet, golden("g18_rvn.npz"), "model_unshared")):
        cfg = meta(z, f"{pre}/cfg")
        model = cls(cfg)
        model.load_state_dict(weights(z, f"{pre}/w/"))
        model = model.to(dev).eval()
        y, S, mask, target = (T(z[f"{pre}/{k}"]).to(dev) for k in ("y", "S", "mask", "target"))
        with torch.no_grad():
            hyb = model(y, S, mask, None, target)
            monkeypatch.setenv("MRIDC_AMD_HYBRID", "0")
            ksp = model(y, S, mask, None, target)
            monkeypatch.delenv("MRIDC_AMD_HYBRID")
        assert_close(hyb, ksp, 5e-5, f"{cls.__name__}: hybrid-space vs k-space")


def test_complex_norm_wrapper_records_gradients():
    """with gradients being recorded the wrapper must not hand back a tensor without grad_fn (its inference path writes raw buffers).
    The recorded path (torch arithmetic, closed-form C^(1/2)) against the oracle's restatement of sensitivity_net.py:17-139 (eigen-decomposition
    form) in float64: the output, and the gradients w.r.t. the input and the regulariser's weights (the reference's mean is a detached scalar)."""
    from mridc_amd.collections.reconstruction.models.sigmanet.sensitivity_net import ComplexNormWrapper
    torch.manual_seed(3)
    reg = torch.nn.Conv2d(2, 2, 3, padding=1)
    wrap = ComplexNormWrapper(reg)
    x = (torch.randn(2, 3, 9, 7, 2) * torch.tensor([1.5, 0.6]) + 0.3).requires_grad_(True)
    out = wrap(x)
    assert out.grad_fn is not None and tuple(out.shape) == tuple(x.shape)
    gout = torch.randn_like(out)
    out.backward(gout)
    xd = x.detach().double().requires_grad_(True)
    regd = torch.nn.Conv2d(2, 2, 3, padding=1).double()
    regd.load_state_dict({k: v.double() for k, v in reg.state_dict().items()})
    mean = xd.detach().mean()
    xx, xy, yx, yy = (c.reshape(-1, 1, 1, 1) for c in oracle.dunet.complex_pseudocovariance_half(xd - mean))
    re, im = torch.unbind(xd - mean, dim=-1)
    det = xx * yy - xy * yx
    z = torch.stack([(yy / det) * re + (-xy / det) * im, (-yx / det) * re + (xx / det) * im], -1).clamp(-6, 6)
    y = regd(z.reshape(6, 9, 7, 2).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(2, 3, 9, 7, 2)
    yr, yi = torch.unbind(y, dim=-1)
    ref = torch.stack([xx * yr + xy * yi, yx * yr + yy * yi], -1) + mean
    ref.backward(gout.double())
    assert_close(out.detach(), ref.detach().float(), 2e-5, "recorded wrapper output")
    assert_close(x.grad, xd.grad.float(), 5e-4, "gradient w.r.t. the input (through the covariance too)")
    assert_close(reg.weight.grad, regd.weight.grad.float(), 5e-4, "gradient w.r.t. the regulariser's weights")
    with torch.no_grad():                                   # inference keeps the kernels (GPU only): on the CPU that path refuses
        with pytest.raises(Exception):
            wrap(x.detach())
