"""CPU: the oracle against the reference-generated golden vectors (G1-G4, G10, G11) and against numpy fp64
(restating the reference's value-pinned tests/collections/reconstruction/test_fft.py:17-199)."""
import numpy as np
import pytest
import torch

import oracle
from tests._util import T, assert_close, assert_exact, meta

NORMS = ["backward", "ortho", "forward", "none"]


def arange_input(shape):
    return torch.from_numpy(np.arange(np.prod(shape)).reshape(shape)).float()


# --- restated reference tests: test_fft.py (np.product -> np.prod) -----------------------------------
@pytest.mark.parametrize("shape", [[3, 3], [4, 6], [10, 8, 4]])
@pytest.mark.parametrize("centered", [True, False])
@pytest.mark.parametrize("norm", ["ortho", "backward", "forward"])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft2_vs_numpy(shape, centered, norm, inverse):
    x = arange_input(shape + [2])
    fn = oracle.fft.ifft2 if inverse else oracle.fft.fft2
    out = oracle.utils.tensor_to_complex_np(fn(x, centered=centered, normalization=norm, spatial_dims=[-2, -1]))
    xin = oracle.utils.tensor_to_complex_np(x)
    if centered:
        xin = np.fft.ifftshift(xin, (-2, -1))
    ref = (np.fft.ifft2 if inverse else np.fft.fft2)(xin, norm=norm)
    if centered:
        ref = np.fft.fftshift(ref, (-2, -1))
    assert np.allclose(out, ref)          # the reference's own criterion (test_fft.py:31)
    # and the definition-based float64 DFT agrees with numpy
    ref2 = oracle.fft.fft2_definition(xin, norm, inverse)
    ref64 = (np.fft.ifft2 if inverse else np.fft.fft2)(xin.astype(np.complex128), norm=norm)
    if centered:
        ref2 = np.fft.fftshift(ref2, (-2, -1))
        ref64 = np.fft.fftshift(ref64, (-2, -1))
    assert np.allclose(ref64, ref2, rtol=1e-9, atol=1e-9 * np.abs(ref64).max())
    assert np.linalg.norm(out - ref64) <= 2e-6 * np.linalg.norm(ref64)


def test_complex_abs_vs_numpy():           # test_fft.py:149-159
    x = arange_input([3, 4, 2])
    assert np.allclose(oracle.utils.complex_abs(x).numpy(), np.abs(oracle.utils.tensor_to_complex_np(x)))


@pytest.mark.parametrize("shift, dim", [(0, 0), (1, 0), (-1, 0), (100, 0), ((1, 2), (1, 2))])
@pytest.mark.parametrize("shape", [[5, 6, 2], [3, 4, 5]])
def test_roll_vs_numpy(shift, dim, shape):  # test_fft.py:162-171
    x = np.arange(np.prod(shape)).reshape(shape)
    s = list(shift) if isinstance(shift, tuple) else [shift]
    dd = list(dim) if isinstance(dim, tuple) else [dim]
    out = oracle.fft.roll(torch.from_numpy(x), s, dd).numpy()
    assert np.array_equal(out, np.roll(x, shift, dim))


@pytest.mark.parametrize("shape", [[5, 3], [2, 4, 6], [7, 1, 5]])
def test_shifts_vs_numpy(shape):            # test_fft.py:174-199
    x = np.arange(np.prod(shape)).reshape(shape)
    assert np.array_equal(oracle.fft.fftshift(torch.from_numpy(x)).numpy(), np.fft.fftshift(x))
    assert np.array_equal(oracle.fft.ifftshift(torch.from_numpy(x)).numpy(), np.fft.ifftshift(x))


# --- golden vectors ---------------------------------------------------------------------------------
def test_g1_fft(golden):
    z = golden("g1_fft.npz")
    for cn in ("a33", "a46", "a1084", "r1318", "r1512", "r1719", "r3124"):
        x = T(z[f"{cn}/x"])
        for c in (0, 1):
            for n in NORMS:
                assert_exact(oracle.fft.fft2(x, bool(c), n, [-2, -1]), T(z[f"{cn}/fft2/c{c}/{n}"]), f"{cn} fft2 c{c} {n}")
                assert_exact(oracle.fft.ifft2(x, bool(c), n, [-2, -1]), T(z[f"{cn}/ifft2/c{c}/{n}"]), f"{cn} ifft2")
    x = T(z["sd/x"])
    assert_exact(oracle.fft.fft2(x, True, "ortho", [-3, -2]), T(z["sd/fft2_m3m2"]))
    assert_exact(oracle.fft.ifft2(x, False, "backward", [1, 2]), T(z["sd/ifft2_12"]))
    xc = torch.view_as_complex(T(z["cplx/x"]))
    assert_exact(oracle.fft.fft2(xc, True, "ortho"), T(z["cplx/fft2"]))


def test_g1_fft_float64_agreement(golden):
    """fp32 torch path vs independent float64 numpy path: norm-relative (appendix C)."""
    z = golden("g1_fft.npz")
    for cn in ("r1318", "r1719", "r3124"):
        x = T(z[f"{cn}/x"])
        xc = oracle.utils.tensor_to_complex_np(x)
        for c in (0, 1):
            for n in NORMS:
                ref = oracle.fft.fft2_np64(xc, bool(c), n)
                got = oracle.utils.tensor_to_complex_np(T(z[f"{cn}/fft2/c{c}/{n}"]))
                assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6


def test_g2_shift(golden):
    z = golden("g2_shift.npz")
    for nm in ("s56", "s732", "s4152", "s9"):
        x = T(z[f"{nm}/x"])
        assert_exact(oracle.fft.fftshift(x), T(z[f"{nm}/fftshift_all"]))
        assert_exact(oracle.fft.ifftshift(x), T(z[f"{nm}/ifftshift_all"]))
        for dim in range(x.dim()):
            assert_exact(oracle.fft.fftshift(x, [dim]), T(z[f"{nm}/fftshift/{dim}"]))
            assert_exact(oracle.fft.ifftshift(x, [dim]), T(z[f"{nm}/ifftshift/{dim}"]))
            for s in (-3, 0, 1, 2, 11):
                assert_exact(oracle.fft.roll(x, [s], [dim]), T(z[f"{nm}/roll/{dim}/{s}"]))
    x = T(z["multi/x"])
    assert_exact(oracle.fft.roll(x, [2, 1], [0, 2]), T(z["multi/roll_0_2"]))
    assert_exact(oracle.fft.fftshift(x, [-2, -1]), T(z["multi/fftshift_m2m1"]))
    assert_exact(oracle.fft.ifftshift(x, [0, 1]), T(z["multi/ifftshift_01"]))
    with pytest.raises(ValueError):
        oracle.fft.roll(x, [1, 2], [0])


def test_g3_complex(golden):
    z = golden("g3_complex.npz")
    x, y, e = T(z["x"]), T(z["y"]), T(z["e"])
    u = oracle.utils
    assert_exact(u.complex_mul(x, y), T(z["complex_mul"]))
    assert_exact(u.complex_mul(e, y), T(z["complex_mul_bcast"]))
    assert_exact(u.complex_conj(x), T(z["complex_conj"]))
    assert_exact(u.complex_abs(x), T(z["complex_abs"]))
    assert_exact(u.complex_abs_sq(x), T(z["complex_abs_sq"]))
    for dim in (0, 1):
        assert_exact(u.rss(x, dim), T(z[f"rss/{dim}"]))
        assert_exact(u.rss_complex(x, dim), T(z[f"rss_complex/{dim}"]))
        assert_exact(u.sense(x, y, dim), T(z[f"sense/{dim}"]))
        assert_exact(u.coil_combination(x, y, "SENSE", dim), T(z[f"cc_sense/{dim}"]))
        assert_exact(u.coil_combination(x, y, "RSS", dim), T(z[f"cc_rss/{dim}"]))
    with pytest.raises(ValueError, match="Output type not supported."):
        u.coil_combination(x, y, "FOO", 0)
    with pytest.raises(ValueError):
        u.complex_mul(x[..., :1], y)
    img = T(z["crop/x"])
    assert_exact(u.center_crop(img, (7, 8)), T(z["crop/center_7_8"]))
    assert_exact(u.center_crop(img, (10, 13)), T(z["crop/center_10_13"]))
    assert_exact(u.complex_center_crop(T(z["crop/cx"]), (6, 9)), T(z["crop/complex_6_9"]))
    with pytest.raises(ValueError, match="Invalid shapes."):
        u.center_crop(img, (12, 3))


def test_g4_llg(golden):
    z = golden("g4_llg.npz")
    eta, S = T(z["eta"]), T(z["S"])
    for i in range(int(z["ncases"])):
        m = meta(z, f"case{i}/meta")
        out = oracle.rim.log_likelihood_gradient(eta, T(z[f"case{i}/y"]), S, T(z[f"case{i}/mask"]), m["sigma"],
                                                 m["centered"], m["norm"], [-2, -1], m["coil_dim"])
        assert_close(out, T(z[f"case{i}/out"]), 1e-6, f"llg case{i} {m}")


def test_g10_ssim(golden):
    z = golden("g10_ssim.npz")
    X, Y, dr = T(z["X"]), T(z["Y"]), T(z["data_range"])
    assert abs(float(oracle.metrics.ssim_loss(X, Y, dr)) - float(z["loss"][0])) < 1e-6
    assert abs(float(oracle.metrics.ssim_loss(X, X, dr)) - float(z["loss_same"][0])) < 1e-6


def test_g11_masks_apply(golden):
    """apply_mask semantics with an existing mask: data*mask + 0.0, bit-exact."""
    z = golden("g11_masks.npz")
    for nm in ("s32x16", "s15x12", "s13x18", "s17x19", "b2"):
        shape = [int(v) for v in z[f"{nm}/shape"]]
        x = arange_input(shape)
        mask = T(z[f"{nm}/mask"])
        outs = [oracle.utils.apply_existing_mask(x[i:i + 1], mask[i:i + 1])[0] for i in range(shape[0])]
        assert_exact(torch.cat(outs), T(z[f"{nm}/masked"]))
