"""SURVEY section 8f rows N1 (per-sample preprocessing on the device) and N2 (bit-identical mask generators).

CPU: the build's mask generators against the reference's masks (g12, bit for bit) and the oracle's restatement of
`MRIDataTransforms.__call__` against the reference's outputs (g13).  GPU: the device-side `MRIDataTransforms` (HIP operators
through the C ABI) against the same goldens and, at the knee size, against the oracle.
"""
import json

import numpy as np
import pytest
import torch

import oracle
from tests._util import T, assert_close, assert_exact


def _mask_funcs():
    from mridc_amd.collections.reconstruction.data import subsample
    return subsample


# ---- N2 -----------------------------------------------------------------------------------------------------------------------
def test_g12_mask_generators_bit_identical(golden):
    sub = _mask_funcs()
    z = golden("g12_mask_generators.npz")
    cases = json.loads(str(z["cases"]))
    assert len(cases) >= 80
    for c in cases:
        fn = sub.create_mask_for_mask_type(c["name"], c["cf"], c["acc"])
        fn.rng.seed(c["rng_seed"])
        seed = tuple(c["seed"]) if isinstance(c["seed"], list) else c["seed"]
        if c["name"].startswith("gaussian"):
            np.random.seed(c["global_seed"])
            m, a = fn(np.array(c["shape"]), seed, c["half"], 0.02)
        else:
            m, a = fn(tuple(c["shape"]), seed)
        assert list(m.shape) == list(z[c["key"] + "/shape"]), c["key"]
        assert m.dtype == torch.float32
        bits = np.packbits(m.numpy().astype(np.uint8).ravel())
        assert np.array_equal(bits, z[c["key"] + "/bits"]), f"mask bits differ: {c}"
        assert float(a) == float(z[c["key"] + "/acc"]), c["key"]


def test_mask_generator_interface():
    sub = _mask_funcs()
    with pytest.raises(ValueError, match="Number of center fractions"):
        sub.RandomMaskFunc([0.08, 0.04], [4])
    with pytest.raises(ValueError, match="3 or more dimensions"):
        sub.RandomMaskFunc([0.08], [4])((10, 2), seed=1)
    with pytest.raises(NotImplementedError, match="not supported"):
        sub.create_mask_for_mask_type("spiral", [0.08], [4])
    fn = sub.RandomMaskFunc([0.08], [4])
    state = fn.rng.get_state()[1].copy()
    a, _ = fn((1, 64, 48, 2), seed=(1, 2, 3))
    b, _ = fn((1, 64, 48, 2), seed=(1, 2, 3))
    assert torch.equal(a, b) and np.array_equal(state, fn.rng.get_state()[1])   # seeded call leaves the generator untouched
    m, acc = sub.Equispaced2DMaskFunc([0.08], [4])((1, 32, 40, 2), seed=5)
    assert m.shape == (1, 32, 40, 1) and acc == 4


# ---- N1: oracle vs the reference's outputs -------------------------------------------------------------------------------------
def _case_inputs(z, c):
    name = c["name"]
    sub = _mask_funcs()
    spec = c["mask"]
    mask_func = None
    if spec is not None:
        specs = spec if isinstance(spec[0], list) else [spec]
        mask_func = [sub.create_mask_for_mask_type(*s_) for s_ in specs]
        if c["mask_as_tuple"]:
            mask_func = tuple(mask_func)
    k, S = z[f"{name}/in/kspace"], z[f"{name}/in/sens"]
    eta = z[f"{name}/in/eta"] if f"{name}/eta" in z.files else np.array([])
    mask_in = [z[f"{name}/in/mask"]] if c["stored_mask"] else None
    kw = dict(c["kwargs"])
    if kw.get("crop_size") is not None:
        kw["crop_size"] = tuple(kw["crop_size"])
    return k, S, eta, mask_in, mask_func, kw


def _check_outputs(z, c, ks, y, Sm, m, e, tgt, acc, tol, what):
    name = c["name"]
    assert_close(ks, T(z[f"{name}/kspace"]), tol, f"{what} {name}: kspace")
    assert_close(Sm, T(z[f"{name}/sens"]), tol, f"{what} {name}: sensitivity maps")
    assert_close(tgt, T(z[f"{name}/target"]), tol, f"{what} {name}: target")
    if f"{name}/eta" in z.files:
        assert_close(e, T(z[f"{name}/eta"]), tol, f"{what} {name}: eta")
    assert isinstance(y, list) == c["list_outputs"], f"{what} {name}: list-valued outputs"
    ys = y if isinstance(y, list) else [y]
    ms = m if isinstance(m, list) else [m]
    accs = acc if isinstance(acc, list) else [acc]
    assert len(ys) == c["n_masks"]
    for i, (yy, mm) in enumerate(zip(ys, ms)):
        assert_close(yy, T(z[f"{name}/y{i}"]), tol, f"{what} {name}: masked k-space {i}")
        ref_m = T(z[f"{name}/mask{i}"])
        assert tuple(mm.shape) == tuple(ref_m.shape) and mm.dtype == ref_m.dtype, f"{what} {name}: mask {i} {mm.shape} {mm.dtype}"
        assert_exact(mm.cpu(), ref_m, f"{what} {name}: mask {i}")
    got_acc = [float(a_.item() if torch.is_tensor(a_) else a_) for a_ in accs]
    assert got_acc == [float(v) for v in z[f"{name}/acc"]], f"{what} {name}: acceleration"


def test_g13_oracle_transforms_vs_reference(golden):
    z = golden("g13_transforms.npz")
    for c in json.loads(str(z["cases"])):
        k, S, eta, mask_in, mask_func, kw = _case_inputs(z, c)
        ks, y, Sm, m, e, tgt, acc = oracle.transforms.preprocess(k, S, mask_in, eta, fname="file_1.h5", mask_func=mask_func, **kw)
        _check_outputs(z, c, ks, y, Sm, m, e, tgt, acc, 2e-6, "oracle")


# ---- N1: the device path -----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_max_abs_and_device_division(dev):
    from mridc_amd import ops
    g = torch.Generator().manual_seed(3)
    for shape in ((7,), (5, 9, 2), (15, 64, 48, 2), (3, 640, 372, 2)):
        x = torch.randn(*shape, generator=g) * 3
        m = ops.max_abs(x.to(dev))
        assert m.shape == () and float(m) == float(x.abs().max())                        # max is exact
        if shape[-1] == 2:
            mc = ops.max_abs(x.to(dev), complex_modulus=True)
            ref = torch.view_as_complex(x).abs().max()
            assert abs(float(mc) - float(ref)) <= 2e-7 * float(ref)                        # modulus: 1 ulp (rounded squares, then the root)
            q = ops.div_by_device_scalar(x.to(dev), mc, modulus=True).cpu()
            assert_close(q, (torch.view_as_complex(x) / ref).abs(), 1e-6, "|x / max|")
        assert_exact(ops.div_by_device_scalar(x.to(dev), m).cpu(), x / x.abs().max(), "x / max")   # IEEE division, same divisor
    xn = torch.tensor([1.0, float("nan"), 2.0])
    assert torch.isnan(ops.max_abs(xn.to(dev)))                                           # torch.max propagates NaN
    with pytest.raises(RuntimeError):
        ops.max_abs(torch.ones(4))                                                         # CPU tensor: no fallback


@pytest.mark.gpu
def test_g13_device_transforms_vs_reference(golden, dev):
    from mridc_amd.collections.reconstruction.parts.transforms import MRIDataTransforms
    z = golden("g13_transforms.npz")
    for c in json.loads(str(z["cases"])):
        k, S, eta, mask_in, mask_func, kw = _case_inputs(z, c)
        t = MRIDataTransforms(mask_func=mask_func, **kw)
        ks, y, Sm, m, e, tgt, fname, sl, acc = t(k, S, mask_in, eta, np.array([]), {}, "file_1.h5", 3)
        assert (fname, sl) == ("file_1.h5", 3)
        for v in [ks, Sm, tgt] + (y if isinstance(y, list) else [y]):
            assert v.device.type == "cuda"
        _check_outputs(z, c, ks.cpu(), [v.cpu() for v in y] if isinstance(y, list) else y.cpu(), Sm.cpu(), m, e.cpu() if e.numel() else e,
                       tgt.cpu(), acc, 1e-5, "device")


@pytest.mark.gpu
def test_device_transforms_knee_size_vs_oracle(dev):
    """15 coils, 640 x 372: the shape of the headline workload; random-1-D mask seeded from the file name like the reference."""
    from mridc_amd.collections.reconstruction.data import subsample
    from mridc_amd.collections.reconstruction.parts.transforms import MRIDataTransforms
    rng = np.random.default_rng(5)
    C, H, W = 15, 640, 372
    k = (rng.standard_normal((C, H, W)) + 1j * rng.standard_normal((C, H, W))).astype(np.complex64)
    S = (rng.standard_normal((C, H, W)) + 1j * rng.standard_normal((C, H, W))).astype(np.complex64)
    kw = dict(normalize_inputs=True, max_norm=True, fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1)
    t = MRIDataTransforms(mask_func=[subsample.RandomMaskFunc([0.08], [4])], **kw)
    ks, y, Sm, m, e, tgt, _, _, acc = t(k, S, None, np.array([]), np.array([]), {}, "file_7.h5", 0)
    rk, ry, rS, rm, re_, rt, racc = oracle.transforms.preprocess(k, S, None, np.array([]), fname="file_7.h5",
                                                                 mask_func=[subsample.RandomMaskFunc([0.08], [4])], **kw)
    assert_close(ks.cpu(), rk, 1e-5, "knee: kspace")
    assert_close(y[0].cpu(), ry[0], 1e-5, "knee: masked kspace")
    assert_close(Sm.cpu(), rS, 1e-6, "knee: maps")
    assert_close(tgt.cpu(), rt, 1e-5, "knee: target")
    assert_exact(m[0].cpu(), rm[0], "knee: mask")
    assert acc == racc == [4]
    # the output feeds the cascades directly: already on the device, contiguous, fp32
    assert y[0].is_contiguous() and y[0].dtype == torch.float32 and y[0].device.type == "cuda"


@pytest.mark.gpu
def test_device_transforms_rejects_unsupported():
    from mridc_amd.collections.reconstruction.parts.transforms import MRIDataTransforms
    for kw in (dict(apply_prewhitening=True), dict(apply_gcc=True), dict(kspace_zero_filling_size=(32, 32)), dict(dimensionality=3)):
        with pytest.raises(NotImplementedError):
            MRIDataTransforms(**kw)


# ---- N3: BaseSensitivityModel --------------------------------------------------------------------------------------------------
def _sens_cases(golden):
    from tests._util import weights
    z = golden("g14_sensnet.npz")
    for nm in json.loads(str(z["names"])):
        cfg = json.loads(str(z[nm + "/cfg"]))
        yield nm, cfg, weights(z, nm + "/w/"), T(z[nm + "/y"]), T(z[nm + "/mask"]), T(z[nm + "/out"])


def test_g14_oracle_sens_net_vs_reference(golden):
    for nm, cfg, p, y, mask, want in _sens_cases(golden):
        got = oracle.models.sens_net_forward(p, cfg, y, mask, cfg["num_low_frequencies"])
        assert_close(got, want, 2e-6, f"oracle sens_net {nm}")


def test_sens_net_state_dict_layout_and_config_keys():
    from mridc_amd import synthetic
    from mridc_amd.collections.reconstruction.models.vn import VarNet
    cfg = dict(synthetic.E2EVN_BASELINE_CFG, use_sens_net=True, sens_chans=4, sens_pools=2, sens_mask_type="2D", sens_normalize=True,
               sens_mask_center=True)
    keys = [k for k in VarNet(cfg).state_dict() if k.startswith("sens_net.")]
    assert "sens_net.norm_unet.unet.down_sample_layers.0.layers.0.weight" in keys and len(keys) == 14
    assert not any(k.startswith("sens_net.") for k in VarNet(dict(synthetic.E2EVN_BASELINE_CFG)).state_dict())


@pytest.mark.gpu
def test_g14_device_sens_net_vs_reference(golden, dev):
    from mridc_amd.collections.reconstruction.models.base import BaseSensitivityModel
    for nm, cfg, p, y, mask, want in _sens_cases(golden):
        net = BaseSensitivityModel(cfg["sens_chans"], cfg["sens_pools"], fft_centered=cfg["fft_centered"],
                                   fft_normalization=cfg["fft_normalization"], spatial_dims=[-2, -1], coil_dim=cfg["coil_dim"],
                                   mask_type=cfg["sens_mask_type"], normalize=cfg["sens_normalize"], mask_center=cfg["sens_mask_center"])
        net.load_state_dict(p)                                        # the reference's key layout, strict
        net = net.to(dev).eval()
        with torch.no_grad():
            got = net(y.to(dev), mask.to(dev), cfg["num_low_frequencies"])
        assert_close(got.cpu(), want, 2e-5, f"device sens_net {nm}")
        if cfg["sens_normalize"]:                                     # unit root-sum-of-squares over coils, exactly what the division is for
            rss = torch.view_as_complex(got).abs().pow(2).sum(1).sqrt()
            assert float((rss - 1).abs().max()) < 1e-5
    with pytest.raises(RuntimeError):
        net.cpu()(y, mask)                                            # no CPU fallback


@pytest.mark.parametrize("shape,acc,calib", [((1, 640, 372, 2), 10, (0.0, 0.0)), ((1, 320, 320, 2), 4, (24.0, 16.0)), ((2, 96, 128, 2), 6, (0.0, 0.0))])
def test_poisson_disc_masks_follow_the_reference_law(shape, acc, calib):
    """Poisson2DMaskFunc (subsample.py:465-633; the CIRIM / qCIRIM YAML default, base_cirim_run.yaml:84-90).  The reference throws its darts on
    Numba's private generator -- not reproducible from a seed even there -- so the masks are pinned by the properties its algorithm
    guarantees: shape / dtype, achieved acceleration within `tol` before the centre disc is added, the calibration block and the centre
    disc fully sampled, the corners r >= 1 empty, every pair of dart-thrown samples outside the other's exclusion ellipse (up to the one
    pixel the integer cell positions lose), sample density falling with the distance from the centre, and reproducibility: the same
    seed gives the same mask and leaves the generator's state untouched."""
    sub = _mask_funcs()
    fn = sub.create_mask_for_mask_type("poisson2d", [0.7], [acc])
    assert isinstance(fn, sub.Poisson2DMaskFunc)
    state = fn.rng.get_state()[1].copy()
    mask, a = fn(shape, seed=(7, 1, 3), calib=calib)
    again, _ = fn(shape, seed=(7, 1, 3), calib=calib)
    other, _ = fn(shape, seed=(7, 1, 4), calib=calib)
    ny, nx = shape[-3], shape[-2]
    assert a == acc and mask.dtype == torch.float32 and tuple(mask.shape) == (1, ny, nx, 1)
    assert torch.equal(mask, again) and not torch.equal(mask, other) and np.array_equal(state, fn.rng.get_state()[1])
    m = mask[0, :, :, 0].numpy().astype(bool)
    disc = fn.centered_circle()
    assert m[disc].all()                                                       # the fully sampled centre disc
    cy0, cy1 = int(ny / 2 - calib[-2] / 2), int(ny / 2 + calib[-2] / 2)
    cx0, cx1 = int(nx / 2 - calib[-1] / 2), int(nx / 2 + calib[-1] / 2)
    assert m[cy0:cy1, cx0:cx1].all()                                           # the calibration block
    darts = m & ~disc
    assert abs(m.size / darts.sum() - acc) < 0.3 + 0.05                          # acceleration (tol = 0.3 on the dart-thrown pattern)
    rows, cols = np.mgrid[:ny, :nx]
    dx = np.maximum(np.abs(cols - nx / 2) - calib[-1] / 2, 0)
    dy = np.maximum(np.abs(rows - ny / 2) - calib[-2] / 2, 0)
    r = np.hypot(dx / dx.max(), dy / dy.max())
    assert not darts[r >= 1].any()                                               # cropped corners
    inner, outer = darts[(r > 0.05) & (r < 0.4)].mean(), darts[(r > 0.6) & (r < 1.0)].mean()
    assert inner > 1.5 * outer                                                   # variable density
    # exclusion: away from the centre samples repel each other -- far fewer touching pairs than independent draws of the same density would
    # give (4 neighbour offsets x cells x p^2); then half-scan zeroes the leading rows
    ring = darts & (r > 0.5)
    ys, xs = np.nonzero(ring)
    touching = 0
    for oy, ox in ((0, 1), (1, 0), (1, 1), (1, -1)):
        yy, xx = ys + oy, xs + ox
        ok = (yy < ny) & (xx >= 0) & (xx < nx)
        touching += int(ring[yy[ok], xx[ok]].sum())
    cells = int(((r > 0.5) & (r < 1)).sum())
    independent = 4.0 * cells * (len(ys) / cells) ** 2
    assert touching <= (0.3 if acc >= 10 else 0.8) * independent, (touching, independent)
    half, _ = fn(shape, seed=(7, 1, 3), calib=calib, half_scan_percentage=0.2)
    assert not half[0, : int(np.round(ny * 0.2))].any() and torch.equal(half[0, int(np.round(ny * 0.2)):], mask[0, int(np.round(ny * 0.2)):])


def test_poisson_mask_that_cannot_reach_the_acceleration_raises():
    sub = _mask_funcs()
    with pytest.raises(ValueError, match="Cannot generate mask"):
        sub.Poisson2DMaskFunc([0.7], [400])((1, 16, 16, 2), seed=1, tol=1e-6)
