#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own leaf modules.

Runs only where /root/reference exists (the build container).  The reference is imported from
where it lies through `_refshim` (no reference source is copied); the outputs are plain data
(.npz: inputs, expected outputs, randomly initialised reference weights, config as JSON).

    python tests/golden/generate_golden.py            # writes tests/golden/g*.npz

Fixture ids follow SURVEY.md section 8c (G1..G11).  Model-level compositions (CIRIM / VarNet /
UNet / ZF) are produced by composing the imported reference blocks exactly as the reference
`forward` bodies do (cirim.py:146-165, vn.py:125-142, unet.py:108-121, zf.py:90-100), because the
model classes subclass a pytorch-lightning base that is not installed here.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

fft = _refshim.load("mridc.collections.common.parts.fft")
utils = _refshim.load("mridc.collections.common.parts.utils")
rim_utils = _refshim.load("mridc.collections.reconstruction.models.rim.rim_utils")
rim_block = _refshim.load("mridc.collections.reconstruction.models.rim.rim_block")
vn_block = _refshim.load("mridc.collections.reconstruction.models.varnet.vn_block")
unet_block = _refshim.load("mridc.collections.reconstruction.models.unet_base.unet_block")
ssim_mod = _refshim.load("mridc.collections.common.losses.ssim")
subsample = _refshim.load("mridc.collections.reconstruction.data.subsample")

torch.set_num_threads(4)


def arange_input(shape):
    # tests/collections/reconstruction/fastmri/conftest.py:16-29 (np.product -> np.prod)
    return torch.from_numpy(np.arange(np.prod(shape)).reshape(shape)).float()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def save(name, d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


def sd(module, prefix="w/"):
    return {prefix + k: v.detach().clone() for k, v in module.state_dict().items()}


NORMS = ["backward", "ortho", "forward", "none"]


def g1_fft():
    d = {}
    cases = {"a33": arange_input([3, 3, 2]), "a46": arange_input([4, 6, 2]), "a1084": arange_input([10, 8, 4, 2]),
             "r1318": rnd([2, 3, 13, 18, 2], 11), "r1512": rnd([1, 2, 15, 12, 2], 12), "r1719": rnd([1, 2, 17, 19, 2], 13),
             "r3124": rnd([1, 2, 31, 24, 2], 14)}
    for cn, x in cases.items():
        d[f"{cn}/x"] = x
        for c in (0, 1):
            for n in NORMS:
                d[f"{cn}/fft2/c{c}/{n}"] = fft.fft2(x, centered=bool(c), normalization=n, spatial_dims=[-2, -1])
                d[f"{cn}/ifft2/c{c}/{n}"] = fft.ifft2(x, centered=bool(c), normalization=n, spatial_dims=[-2, -1])
    # spatial_dims other than the last two (interpreted on the complex view, fft.py:66-72)
    x = rnd([2, 6, 5, 4, 2], 15)
    d["sd/x"] = x
    d["sd/fft2_m3m2"] = fft.fft2(x, centered=True, normalization="ortho", spatial_dims=[-3, -2])
    d["sd/ifft2_12"] = fft.ifft2(x, centered=False, normalization="backward", spatial_dims=[1, 2])
    # already-complex input (appendix D.18): last dim != 2 -> no view_as_complex, real view returned
    xc = torch.view_as_complex(rnd([2, 7, 6, 2], 16))
    d["cplx/x"] = torch.view_as_real(xc)
    d["cplx/fft2"] = fft.fft2(xc, centered=True, normalization="ortho")
    # large case: strided sample + norms only (keeps the fixture small)
    xl = rnd([1, 15, 640, 372, 2], 17)
    for c, n in ((0, "backward"), (1, "ortho")):
        for nm, fn in (("fft2", fft.fft2), ("ifft2", fft.ifft2)):
            y = fn(xl, centered=bool(c), normalization=n).reshape(-1)
            d[f"big/{nm}/c{c}/{n}/sample"] = y[::9973].clone()
            d[f"big/{nm}/c{c}/{n}/l2"] = y.double().norm().reshape(1)
    d["big/seed_shape"] = np.array([17, 1, 15, 640, 372, 2])
    save("g1_fft.npz", d)


def g2_shift():
    d = {}
    for nm, shape in (("s56", [5, 6]), ("s732", [7, 3, 2]), ("s4152", [4, 1, 5, 2]), ("s9", [9])):
        x = arange_input(shape)
        d[f"{nm}/x"] = x
        d[f"{nm}/fftshift_all"] = fft.fftshift(x)
        d[f"{nm}/ifftshift_all"] = fft.ifftshift(x)
        for dim in range(len(shape)):
            d[f"{nm}/fftshift/{dim}"] = fft.fftshift(x, dim=[dim])
            d[f"{nm}/ifftshift/{dim}"] = fft.ifftshift(x, dim=[dim])
            for s in (-3, 0, 1, 2, 11):
                d[f"{nm}/roll/{dim}/{s}"] = fft.roll(x, [s], [dim])
    x = arange_input([5, 6, 3])
    d["multi/x"] = x
    d["multi/roll_0_2"] = fft.roll(x, [2, 1], [0, 2])
    d["multi/fftshift_m2m1"] = fft.fftshift(x, dim=[-2, -1])
    d["multi/ifftshift_01"] = fft.ifftshift(x, dim=[0, 1])
    save("g2_shift.npz", d)


def g3_complex():
    d = {}
    x = rnd([2, 4, 9, 7, 2], 21)
    y = rnd([2, 4, 9, 7, 2], 22)
    e = rnd([2, 1, 9, 7, 2], 23)
    d.update(x=x, y=y, e=e)
    d["complex_mul"] = utils.complex_mul(x, y)
    d["complex_mul_bcast"] = utils.complex_mul(e, y)
    d["complex_conj"] = utils.complex_conj(x)
    d["complex_abs"] = utils.complex_abs(x)
    d["complex_abs_sq"] = utils.complex_abs_sq(x)
    for dim in (0, 1):
        d[f"rss/{dim}"] = utils.rss(x, dim)
        d[f"rss_complex/{dim}"] = utils.rss_complex(x, dim)
        d[f"sense/{dim}"] = utils.sense(x, y, dim)
        d[f"cc_sense/{dim}"] = utils.coil_combination(x, y, "SENSE", dim)
        d[f"cc_rss/{dim}"] = utils.coil_combination(x, y, "RSS", dim)
    img = rnd([2, 3, 11, 14], 24)
    d["crop/x"] = img
    d["crop/center_7_8"] = utils.center_crop(img, (7, 8))
    d["crop/center_10_13"] = utils.center_crop(img, (10, 13))
    cimg = rnd([2, 11, 14, 2], 25)
    d["crop/cx"] = cimg
    d["crop/complex_6_9"] = utils.complex_center_crop(cimg, (6, 9))
    a, b = utils.center_crop_to_smallest(rnd([2, 9, 14], 26), img[:, 0])
    d["crop/smallest_a"], d["crop/smallest_b"] = a, b
    save("g3_complex.npz", d)


def make_mask(shape, seed=123, cf=0.08, acc=4):
    """RandomMaskFunc through apply_mask exactly as the reference model tests do (test_cirim.py:307-318)."""
    mf = subsample.RandomMaskFunc([cf], [acc])
    x = arange_input(shape)
    outs, masks = [], []
    for i in range(x.shape[0]):
        o, m, _ = utils.apply_mask(x[i: i + 1], mf, seed=seed)
        outs.append(o)
        masks.append(m)
    return torch.cat(outs), torch.cat(masks)


def g11_masks():
    d = {}
    for nm, shape in (("s32x16", [1, 3, 32, 16, 2]), ("s15x12", [1, 5, 15, 12, 2]), ("s13x18", [1, 8, 13, 18, 2]),
                      ("s17x19", [1, 2, 17, 19, 2]), ("b2", [2, 3, 12, 10, 2])):
        o, m = make_mask(shape)
        d[f"{nm}/shape"] = np.array(shape)
        d[f"{nm}/masked"] = o
        d[f"{nm}/mask"] = m
    mf = subsample.RandomMaskFunc([0.08], [4])
    m, acc = mf([1, 640, 372, 2], seed=123)
    d["knee/mask_372"] = m
    mf = subsample.Equispaced1DMaskFunc([0.08], [4])
    m, acc = mf([1, 640, 372, 2], seed=123)
    d["knee/equi_372"] = m
    save("g11_masks.npz", d)


def g12_mask_generators():
    """N2: every reproducible mask generator of the reference, bit for bit (masks stored packed, 1 bit per sample).

    Random / Equispaced draw from the object's own RandomState under `temp_seed`; the Gaussian ones use the global np.random
    (subsample.py:353,438), so the generator seeds that explicitly -- the test does the same before calling the build's functions.
    """
    d, cases = {}, []
    combos = {"random1d": [([0.08], [4]), ([0.08, 0.04], [4, 8]), ([0.04], [8])],
              "equispaced1d": [([0.08], [4]), ([0.08, 0.04], [4, 8]), ([0.04], [8])],
              "equispaced2d": [([0.08], [4]), ([0.04], [8]), ([0.7], [10])],
              "gaussian1d": [([0.7], [4]), ([0.7, 0.5], [4, 8])],
              "gaussian2d": [([0.7], [10]), ([0.5], [4])]}
    shapes = [(1, 640, 372, 2), (15, 64, 48, 2), (3, 17, 19, 2), (1, 320, 320, 2)]
    seeds = [123, (102, 105, 108, 101, 95, 49, 46, 104, 53)]       # an int and tuple(map(ord, "file_1.h5"))
    for name, cfs in combos.items():
        for ci, (cf, acc) in enumerate(cfs):
            for si, shape in enumerate(shapes):
                if name.startswith("gaussian") and shape[1] == 640:
                    continue                                            # np.random.choice over 238 080 cells: slow and adds nothing
                for ki, seed in enumerate(seeds):
                    fn = subsample.create_mask_for_mask_type(name, cf, acc)
                    fn.rng.seed(2024)                                   # state of the object's generator outside temp_seed
                    half = 0.25 if (name.startswith("gaussian") and si == 1) else 0.0
                    if name.startswith("gaussian"):
                        np.random.seed(4321 + ki)
                        m, a = fn(np.array(shape), seed, half, 0.02)
                    else:
                        m, a = fn(shape, seed)
                    key = f"{name}/{ci}/{si}/{ki}"
                    d[key + "/bits"] = np.packbits(m.numpy().astype(np.uint8).ravel())
                    d[key + "/shape"] = np.array(m.shape)
                    d[key + "/acc"] = np.array(float(a))
                    cases.append(dict(key=key, name=name, cf=cf, acc=acc, shape=list(shape), seed=seed, half=half,
                                      global_seed=4321 + ki, rng_seed=2024))
    d["cases"] = np.array(json.dumps(cases))
    save("g12_mask_generators.npz", d)


def g13_transforms():
    """N1: the reference's MRIDataTransforms on small synthetic slices (inputs, constructor kwargs, every tensor output)."""
    T = _refshim.load("mridc.collections.reconstruction.parts.transforms")
    rng = np.random.default_rng(77)

    def cplx(*shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)

    base = dict(fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, use_seed=True)
    cases = [
        ("sense_norm", dict(base, normalize_inputs=True, max_norm=True, mask=("random1d", [0.08], [4])), (4, 16, 12), None),
        ("rss_ortho_centered", dict(base, coil_combination_method="RSS", fft_centered=True, fft_normalization="ortho",
                                    normalize_inputs=True, mask=("equispaced1d", [0.08], [4])), (5, 17, 19), None),
        ("crop_before", dict(base, crop_size=(12, 10), normalize_inputs=True, mask=("random1d", [0.08], [4])), (4, 20, 16), None),
        ("crop_after_kspace", dict(base, crop_size=(12, 10), kspace_crop=True, crop_before_masking=False, normalize_inputs=True,
                                   mask=("equispaced2d", [0.08], [4]), mask_as_tuple=True), (4, 20, 16), None),
        ("stored_mask_shift", dict(base, shift_mask=True, normalize_inputs=True, fft_normalization="forward"), (3, 14, 18), "stored"),
        ("none_norm", dict(base, fft_normalization="none", normalize_inputs=True, mask=("equispaced2d", [0.08], [4])), (4, 16, 16), None),
        ("two_masks", dict(base, normalize_inputs=True, mask=[("random1d", [0.08], [4]), ("equispaced1d", [0.04], [8])]),
         (3, 16, 24), None),
        ("fully_sampled", dict(base, normalize_inputs=False), (3, 10, 12), None),
    ]
    d, meta = {}, []
    for name, kw, shape, stored in cases:
        kw = dict(kw)
        spec = kw.pop("mask", None)
        as_tuple = kw.pop("mask_as_tuple", False)   # a non-list container takes the single-mask branch (transforms.py:468-478)
        mask_func = None
        if spec is not None:
            specs = spec if isinstance(spec, list) else [spec]
            mask_func = [subsample.create_mask_for_mask_type(*s_) for s_ in specs]
            if as_tuple:
                mask_func = tuple(mask_func)
        k, S, eta = cplx(*shape), cplx(*shape), cplx(*shape[1:])
        mask_in = None
        if stored:
            mask_in = [(rng.random(shape[1:]) < 0.4).astype(np.float32)]
        t = T.MRIDataTransforms(mask_func=mask_func, **kw)
        out = t(k, S, mask_in, eta if name != "fully_sampled" else np.array([]), np.array([]), {}, "file_1.h5", 3)
        ks, y, Sm, m, e, tgt, _, _, acc = out
        d[f"{name}/in/kspace"], d[f"{name}/in/sens"], d[f"{name}/in/eta"] = k, S, eta
        if mask_in is not None:
            d[f"{name}/in/mask"] = mask_in[0]
        d[f"{name}/kspace"], d[f"{name}/sens"], d[f"{name}/target"] = ks, Sm, tgt
        if e is not None and torch.is_tensor(e) and e.numel():
            d[f"{name}/eta"] = e
        ys = y if isinstance(y, list) else [y]
        ms = m if isinstance(m, list) else [m]
        for i, (yy, mm) in enumerate(zip(ys, ms)):
            d[f"{name}/y{i}"] = yy
            d[f"{name}/mask{i}"] = mm
        accs = acc if isinstance(acc, list) else [acc]
        d[f"{name}/acc"] = np.array([float(a if not torch.is_tensor(a) else a.item()) for a in accs])
        meta.append(dict(name=name, kwargs={k_: (list(v) if isinstance(v, tuple) else v) for k_, v in kw.items()},
                         mask=spec, mask_as_tuple=as_tuple, shape=list(shape), stored_mask=bool(stored), n_masks=len(ys),
                         list_outputs=isinstance(y, list)))
    d["cases"] = np.array(json.dumps(meta))
    save("g13_transforms.npz", d)


def g14_sensnet():
    """N3: BaseSensitivityModel.  models/base.py imports pytorch-lightning, so the forward body (base.py:886-932) is composed from
    the imported reference pieces exactly as those lines do: get_pad_and_num_low_freqs (:842-884, restated line by line below),
    utils.batched_mask_center, fft.ifft2, unet_block.NormUnet on the coils-as-batch view, utils.rss_complex."""
    def pad_and_nlf(mask, num_low_frequencies=None):
        if num_low_frequencies is None or num_low_frequencies == 0:
            squeezed_mask = mask[:, 0, 0, :, 0].to(torch.int8)
            cent = torch.div(squeezed_mask.shape[1], 2, rounding_mode="trunc")
            left = torch.argmin(squeezed_mask[:, :cent].flip(1), dim=1)
            right = torch.argmin(squeezed_mask[:, cent:], dim=1)
            nlf = torch.max(2 * torch.min(left, right), torch.ones_like(left))
        else:
            nlf = num_low_frequencies * torch.ones(mask.shape[0], dtype=mask.dtype, device=mask.device)
        return torch.div(mask.shape[-2] - nlf + 1, 2, rounding_mode="trunc"), nlf

    cases = [("default", dict(sens_chans=4, sens_pools=2, sens_mask_type="2D", sens_normalize=True, sens_mask_center=True,
                              fft_centered=False, fft_normalization="backward", coil_dim=1), (1, 4, 20, 24), None, 1),
             ("ortho_1d_nlf", dict(sens_chans=4, sens_pools=2, sens_mask_type="1D", sens_normalize=True, sens_mask_center=True,
                                   fft_centered=True, fft_normalization="ortho", coil_dim=1), (1, 3, 17, 18), 4, 1),
             ("no_center_no_norm", dict(sens_chans=6, sens_pools=1, sens_mask_type="2D", sens_normalize=False, sens_mask_center=False,
                                        fft_centered=False, fft_normalization="backward", coil_dim=1), (1, 5, 12, 16), None, 1),
             ("per_batch_masks", dict(sens_chans=4, sens_pools=2, sens_mask_type="2D", sens_normalize=True, sens_mask_center=True,
                                      fft_centered=False, fft_normalization="backward", coil_dim=1), (2, 3, 16, 20), None, 2)]
    d = {}
    for i, (nm, cfg, shape, nlf_arg, mask_batch) in enumerate(cases):
        B, C, H, W = shape
        torch.manual_seed(700 + i)
        net = unet_block.NormUnet(cfg["sens_chans"], cfg["sens_pools"], in_chans=2, out_chans=2, drop_prob=0.0, padding_size=15,
                                  normalize=cfg["sens_normalize"])
        k = rnd([B, C, H, W, 2], 710 + i)
        masks = []
        for b_ in range(mask_batch):
            mf = subsample.RandomMaskFunc([0.2 + 0.1 * b_], [2])
            m, _ = mf([1, H, W, 2], seed=11 + b_)
            masks.append(m.reshape(1, 1, 1, W, 1))
        mask = torch.cat(masks, 0)
        y = k * mask
        with torch.no_grad():
            x = y
            if cfg["sens_mask_center"]:
                pad, nlf = pad_and_nlf(mask, nlf_arg)
                x = utils.batched_mask_center(x, pad, pad + nlf, mask_type=cfg["sens_mask_type"])
            img = fft.ifft2(x, centered=cfg["fft_centered"], normalization=cfg["fft_normalization"], spatial_dims=[-2, -1])
            b, c, h, w, comp = img.shape
            out = net(img.view(b * c, 1, h, w, comp))
            out = out.view(b, c, h, w, comp)
            if cfg["sens_normalize"]:
                out = out / utils.rss_complex(out, dim=cfg["coil_dim"]).unsqueeze(-1).unsqueeze(cfg["coil_dim"])
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(cfg, num_low_frequencies=nlf_arg)))
        d[f"{nm}/y"], d[f"{nm}/mask"], d[f"{nm}/out"] = y, mask, out
        d.update(sd(net, f"{nm}/w/norm_unet."))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g14_sensnet.npz", d)


def synth(B, C, H, W, seed):
    """Small smooth-ish multicoil problem: image, sens maps (sum |S|^2 = 1), full k-space (centred ortho)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 1, H, W, 2, generator=g)
    S = torch.randn(B, C, H, W, 2, generator=g)
    S = S / utils.complex_abs_sq(S).sum(1, keepdim=True).sqrt().unsqueeze(-1)
    return img, S


def g4_llg():
    d = {}
    B, C, H, W = 2, 3, 12, 10
    img, S = synth(B, C, H, W, 41)
    eta = rnd([B, H, W, 2], 42)
    d.update(eta=eta, S=S)
    i = 0
    for centered, norm in ((True, "ortho"), (False, "backward"), (True, "forward"), (False, "none")):
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m1 = make_mask([1, C, H, W, 2])            # [1,1,1,W,1] fp32
        mB = torch.cat([m1, torch.roll(m1, 3, dims=-2)], 0)   # [B,1,1,W,1]
        g2 = torch.Generator().manual_seed(43)
        m2d = (torch.rand(1, 1, H, W, 1, generator=g2) < 0.4).float()
        for mname, m in (("m1", m1), ("mB", mB), ("m2d", m2d)):
            y = k * m
            for dt in ("bool", "uint8", "float32"):
                mm = m.to(getattr(torch, dt))
                for cd in (0, 1):
                    for sigma in (1.0, 0.5):
                        if sigma != 1.0 and (dt != "float32" or cd != 1):
                            continue
                        key = f"case{i}"
                        out = rim_utils.log_likelihood_gradient(eta, y, S, mm, sigma, centered, norm, [-2, -1], cd)
                        d[key + "/y"] = y
                        d[key + "/mask"] = mm
                        d[key + "/out"] = out
                        d[key + "/meta"] = np.array(json.dumps(dict(centered=centered, norm=norm, mask=mname, dtype=dt,
                                                                   coil_dim=cd, sigma=sigma)))
                        i += 1
    d["ncases"] = np.array(i)
    # big single-step checksum at the headline size
    img, S = synth(1, 15, 640, 372, 44)
    eta = rnd([1, 640, 372, 2], 45, 0.1)
    k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
    mk = subsample.RandomMaskFunc([0.08], [4])([1, 640, 372, 2], seed=123)[0].reshape(1, 1, 1, 372, 1).bool()
    out = rim_utils.log_likelihood_gradient(eta, k * mk, S, mk, 1.0, False, "backward", [-2, -1], 1).reshape(-1)
    d["big/sample"] = out[::4999].clone()
    d["big/l2"] = out.double().norm().reshape(1)
    save("g4_llg.npz", d)


RIM_CFG = dict(recurrent_layer="IndRNN", conv_filters=[64, 64, 2], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1],
               conv_bias=[True, True, False], recurrent_filters=[64, 64, 0], recurrent_kernels=[1, 1, 0],
               recurrent_dilations=[1, 1, 0], recurrent_bias=[True, True, False], depth=2, time_steps=8, conv_dim=2,
               no_dc=True, fft_centered=True, fft_normalization="ortho", spatial_dims=[-2, -1], coil_dim=1,
               dimensionality=2)


def scale_weights(mod, factor):
    # reference init makes the IndRNN net nearly linear (ih / hh std = 1/(hid*(1+k^2)), rnn_cells.py:306,319;
    # SURVEY appendix C) -> scale those two so the ReLUs and the recurrence actually shape the output
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("rnn.ih.weight") or n.endswith("rnn.hh"):
                p.mul_(factor)


def g5_rimblock():
    d = {}
    cases = []
    # (name, cfg overrides, shape, keep_eta, pred_is_list, weight scale)
    cases.append(("ind64", dict(time_steps=3), [1, 3, 16, 12, 2], False, False, 1.0))
    cases.append(("ind64_keep", dict(time_steps=2, fft_centered=False, fft_normalization="backward"), [1, 3, 16, 12, 2], True, False, 8.0))
    cases.append(("ind16_dc", dict(time_steps=3, no_dc=False, conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0]), [2, 4, 13, 18, 2], False, False, 5.0))
    cases.append(("ind16_list", dict(time_steps=2, conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0]), [1, 5, 15, 12, 2], True, True, 5.0))
    cases.append(("gru16", dict(time_steps=3, recurrent_layer="GRU", conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0], recurrent_kernels=[3, 3, 0]), [1, 3, 16, 12, 2], False, False, 1.0))
    cases.append(("gru8_k1", dict(time_steps=2, recurrent_layer="GRU", conv_filters=[8, 8, 2], recurrent_filters=[8, 8, 0]), [2, 2, 9, 11, 2], False, False, 1.0))
    cases.append(("mgu16", dict(time_steps=3, recurrent_layer="MGU", conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0], recurrent_kernels=[3, 3, 0], recurrent_dilations=[1, 2, 0]), [1, 3, 16, 12, 2], False, False, 1.0))
    for i, (nm, ov, shape, keep, aslist, wscale) in enumerate(cases):
        cfg = dict(RIM_CFG)
        cfg.update(ov)
        torch.manual_seed(500 + i)
        blk = rim_block.RIMBlock(**cfg).eval()
        scale_weights(blk, wscale)
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 510 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=cfg["fft_centered"], normalization=cfg["fft_normalization"])
        _, m = make_mask([1, C, H, W, 2])
        m = m.bool()
        y = k * m
        if keep:
            p0 = utils.sense(fft.ifft2(y, centered=cfg["fft_centered"], normalization=cfg["fft_normalization"]), S, 1)
            pred = [p0 * 0.5, p0] if aslist else p0
        else:
            pred = y
        with torch.no_grad():
            outs, hx = blk(pred, y, S, m, None, None, 1.0, keep_eta=keep)
        d[f"{nm}/cfg"] = np.array(json.dumps(cfg))
        d[f"{nm}/meta"] = np.array(json.dumps(dict(keep_eta=keep, pred_is_list=aslist)))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = y, S, m
        if keep:
            d[f"{nm}/pred"] = p0
        d[f"{nm}/outs"] = torch.stack(outs)
        for j, h in enumerate(hx):
            d[f"{nm}/hx{j}"] = h
        d.update(sd(blk, f"{nm}/w/"))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g5_rimblock.npz", d)


def compose_cirim(blocks, cfg, y, S, mask, init_pred, target):
    # restates cirim.py:146-165 + process_intermediate_pred :187-197 over imported RIMBlocks
    prediction = y.clone()
    init_pred = None if init_pred is None or init_pred.dim() < 4 else init_pred
    cascades = []
    for i, cascade in enumerate(blocks):
        prediction, _ = cascade(prediction, y, S, mask, init_pred, None, 1.0,
                                keep_eta=False if i == 0 else cfg["keep_eta"])
        steps = []
        for pred in prediction:
            if not cfg["no_dc"]:
                pred = fft.ifft2(pred, centered=cfg["fft_centered"], normalization=cfg["fft_normalization"],
                                 spatial_dims=cfg["spatial_dims"])
                pred = utils.coil_combination(pred, S, method=cfg["coil_combination_method"], dim=cfg["coil_dim"])
            pred = torch.view_as_complex(pred)
            _, pred = utils.center_crop_to_smallest(target, pred)
            steps.append(pred)
        cascades.append(steps)
    return cascades


def g6_cirim():
    import math
    d = {}
    cases = [
        ("c8f16", dict(conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0], num_cascades=8, time_steps=5,
                       fft_centered=False, fft_normalization="backward", keep_eta=True, no_dc=True),
         [1, 6, 32, 24, 2], 3.0, (28, 20)),
        ("c2f64", dict(num_cascades=2, time_steps=8, keep_eta=True, no_dc=True), [1, 4, 24, 20, 2], 6.0, None),
        ("c2f16dc", dict(conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0], num_cascades=2, time_steps=8,
                         keep_eta=False, no_dc=False), [2, 3, 15, 12, 2], 4.0, None),
    ]
    for i, (nm, ov, shape, wscale, crop) in enumerate(cases):
        cfg = dict(RIM_CFG)
        cfg.update(coil_combination_method="SENSE", keep_eta=True, num_cascades=1)
        cfg.update(ov)
        T = 8 * math.ceil(cfg["time_steps"] / 8)       # cirim.py:51
        bcfg = {k: v for k, v in cfg.items() if k not in ("coil_combination_method", "keep_eta", "num_cascades")}
        bcfg["time_steps"] = T
        torch.manual_seed(600 + i)
        blocks = [rim_block.RIMBlock(**bcfg).eval() for _ in range(cfg["num_cascades"])]
        for b in blocks:
            scale_weights(b, wscale)
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 610 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=cfg["fft_centered"], normalization=cfg["fft_normalization"])
        _, m = make_mask([1, C, H, W, 2])
        m = m.bool()
        y = k * m
        target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=cfg["fft_centered"],
                                                         normalization=cfg["fft_normalization"]), S, 1))
        if crop is not None:
            target = utils.center_crop(target, crop)
        with torch.no_grad():
            out = compose_cirim(blocks, cfg, y, S, m, None, target)
        d[f"{nm}/cfg"] = np.array(json.dumps(cfg))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"], d[f"{nm}/target"] = y, S, m, target
        d[f"{nm}/out"] = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
        for ci, b in enumerate(blocks):
            d.update(sd(b, f"{nm}/w/cirim.{ci}."))
        # loss with the reference's weighting (cirim.py:218-247), l1
        l1 = torch.nn.L1Loss()
        tgt = torch.abs(target / torch.max(torch.abs(target)))
        closs = []
        for cp in out:
            ls = [l1(tgt, torch.abs(t / torch.max(torch.abs(t)))) for t in cp]
            _l = [x * torch.logspace(-1, 0, steps=T).to(ls[0]) for x in ls]
            closs.append(sum(sum(_l) / T))
        d[f"{nm}/loss_l1"] = (sum(closs) / len(blocks)).reshape(1)
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g6_cirim.npz", d)


def g19_cirim_spec():
    """G6 exactly as SURVEY 8c specifies it -- the model-zoo CIRIM (8 cascades x time_steps 5 -> 8, IndRNN, 64 filters, base_cirim_run.yaml)
    at [1,15,64,48,2], the full 64-step chain -- plus the row-H harness outputs of the final estimate: the `abs / max` image
    (models/base.py:415-419) and MSE / NMSE / PSNR / SSIM with maxval = output.max() - output.min() (base.py:427-436; formulas of
    common/metrics/reconstruction_metrics.py:11-41, SSIM through the reference's own SSIMLoss: skimage's 7x7 uniform-window SSIM with
    the sample covariance and the border cropped is 1 - SSIMLoss on the valid region)."""
    import math
    d = {}
    cfg = dict(RIM_CFG)
    cfg.update(coil_combination_method="SENSE", keep_eta=True, num_cascades=8, time_steps=5, fft_centered=False,
               fft_normalization="backward", no_dc=True)
    T = 8 * math.ceil(cfg["time_steps"] / 8)
    bcfg = {k: v for k, v in cfg.items() if k not in ("coil_combination_method", "keep_eta", "num_cascades")}
    bcfg["time_steps"] = T
    torch.manual_seed(1900)
    blocks = [rim_block.RIMBlock(**bcfg).eval() for _ in range(cfg["num_cascades"])]
    for b in blocks:
        scale_weights(b, 4.0)
    B, C, H, W = 1, 15, 64, 48
    img, S = synth(B, C, H, W, 1910)
    k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
    _, m = make_mask([1, C, H, W, 2])
    m = m.bool()
    y = k * m
    target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=False, normalization="backward"), S, 1))
    with torch.no_grad():
        out = compose_cirim(blocks, cfg, y, S, m, None, target)
    d["cfg"] = np.array(json.dumps(cfg))
    d["y"], d["S"], d["mask"], d["target"] = y, S, m, target
    d["out"] = torch.view_as_real(torch.stack([torch.stack(c) for c in out]))
    for ci, b in enumerate(blocks):
        d.update(sd(b, f"w/cirim.{ci}."))
    # harness post-processing + metrics of the final estimate (test_step, base.py:394-436)
    preds = out[-1][-1]
    output = torch.abs(preds).detach().cpu()
    output = output / output.max()
    tgt = torch.abs(target).detach().cpu()
    tgt = tgt / tgt.max()
    o, t = output.numpy(), tgt.numpy()
    maxval = o.max() - o.min()
    mse = np.mean((t - o) ** 2)
    nmse = np.linalg.norm(t - o) ** 2 / np.linalg.norm(t) ** 2
    psnr = 10 * np.log10(float(maxval) ** 2 / np.mean((t.astype(np.float64) - o.astype(np.float64)) ** 2))
    loss = ssim_mod.SSIMLoss().double()
    ssim = 1.0 - float(loss(tgt[:, None].double(), output[:, None].double(), torch.tensor([float(maxval)], dtype=torch.float64)))
    d["harness/output"], d["harness/target"] = output, tgt
    d["harness/metrics"] = np.array([mse, nmse, ssim, psnr, maxval], dtype=np.float64)     # MSE, NMSE, SSIM, PSNR, maxval
    save("g19_cirim_spec.npz", d)


def g20_rim3d():
    """A14, 3-D mode of RIMBlock (rim_block.py:168-180,230-246; the cases of tests/collections/reconstruction/models/test_cirim.py:155-290):
    [batch, slices, coils, H, W, 2] inputs, Conv3d / ReplicationPad3d layers over (slices folded with batch, H, W)."""
    d = {}
    cases = [("s1", [1, 1, 3, 15, 12, 2], dict(time_steps=2), 5.0), ("b3s2", [3, 2, 5, 15, 12, 2], dict(time_steps=2), 5.0),
             ("s2_16", [1, 2, 4, 12, 10, 2], dict(time_steps=3, conv_filters=[16, 16, 2], recurrent_filters=[16, 16, 0], fft_centered=False,
                                                  fft_normalization="backward"), 5.0)]
    for i, (nm, shape, ov, wscale) in enumerate(cases):
        cfg = dict(RIM_CFG)
        cfg.update(conv_dim=3, dimensionality=3)
        cfg.update(ov)
        torch.manual_seed(2000 + i)
        blk = rim_block.RIMBlock(**cfg).eval()
        scale_weights(blk, wscale)
        B, S, C, H, W, _ = shape
        img, Smap = synth(B * S, C, H, W, 2010 + i)
        k = fft.fft2(utils.complex_mul(img, Smap), centered=cfg["fft_centered"], normalization=cfg["fft_normalization"])
        _, m = make_mask([1, C, H, W, 2])
        y = (k * m).reshape(B, S, C, H, W, 2)
        Smap = Smap.reshape(B, S, C, H, W, 2)
        mask = m.reshape(1, 1, 1, 1, W, 1).expand(B, S, 1, 1, W, 1).contiguous()
        with torch.no_grad():
            outs, hx = blk(y, y, Smap, mask, None, None, 1.0, keep_eta=False)
        d[f"{nm}/cfg"] = np.array(json.dumps(cfg))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = y, Smap, mask
        d[f"{nm}/outs"] = torch.stack(outs)
        for j, h in enumerate(hx):
            d[f"{nm}/hx{j}"] = h
        d.update(sd(blk, f"{nm}/w/"))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g20_rim3d.npz", d)


def g7_varnet():
    d = {}
    cases = [("u14p2", 14, 2, 11, [1, 3, 32, 16, 2], True, False), ("u14p2_odd", 14, 2, 11, [1, 5, 15, 12, 2], False, False),
             ("u4p4", 4, 4, 15, [1, 2, 17, 19, 2], True, True), ("u6p3_nonorm", 6, 3, 7, [2, 2, 21, 26, 2], False, False)]
    for i, (nm, ch, pools, pad, shape, centered, no_dc) in enumerate(cases):
        torch.manual_seed(700 + i)
        normalize = "nonorm" not in nm
        nu = unet_block.NormUnet(ch, pools, padding_size=pad, normalize=normalize)
        norm = "ortho" if centered else "backward"
        blk = vn_block.VarNetBlock(nu, fft_centered=centered, fft_normalization=norm, spatial_dims=[-2, -1],
                                   coil_dim=1, no_dc=no_dc).eval()
        with torch.no_grad():
            blk.dc_weight.fill_(0.7)
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 710 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        pred = y + 0.1 * rnd(list(y.shape), 720 + i)
        mm = m.byte() if i % 2 == 0 else m.bool()       # pipeline emits uint8 (transforms.py:467)
        with torch.no_grad():
            eta_in = blk.sens_reduce(pred, S)
            nu_out = nu(eta_in)
            out = blk(pred, y, S, mm)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(chans=ch, num_pools=pools, padding_size=pad, normalize=normalize,
                                                  fft_centered=centered, fft_normalization=norm, no_dc=no_dc)))
        d[f"{nm}/pred"], d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = pred, y, S, mm
        d[f"{nm}/eta_in"], d[f"{nm}/normunet_out"], d[f"{nm}/out"] = eta_in, nu_out, out
        d.update(sd(blk, f"{nm}/w/"))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g7_varnet.npz", d)


def g8_models():
    d = {}
    # VarNet: 3 cascades (ch 8, pools 2, pad 11), vn.py:125-142
    torch.manual_seed(800)
    cfg = dict(num_cascades=3, channels=8, pooling_layers=2, padding_size=11, normalize=True, no_dc=False,
               fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1,
               coil_combination_method="SENSE")
    blocks = [vn_block.VarNetBlock(unet_block.NormUnet(8, 2, padding_size=11, normalize=True), fft_centered=False,
                                   fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, no_dc=False).eval()
              for _ in range(3)]
    B, C, H, W = 1, 4, 30, 22
    img, S = synth(B, C, H, W, 810)
    k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
    _, m = make_mask([1, C, H, W, 2])
    m = m.byte()
    y = k * m
    target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=False, normalization="backward"), S, 1))
    target = utils.center_crop(target, (26, 20))
    with torch.no_grad():
        est = y.clone()
        for b in blocks:
            est = b(est, y, S, m)
        est = fft.ifft2(est, centered=False, normalization="backward", spatial_dims=[-2, -1])
        est = utils.coil_combination(est, S, method="SENSE", dim=1)
        est = torch.view_as_complex(est)
        _, est = utils.center_crop_to_smallest(target, est)
    d["vn/cfg"] = np.array(json.dumps(cfg))
    d["vn/y"], d["vn/S"], d["vn/mask"], d["vn/target"] = y, S, m, target
    d["vn/out"] = torch.view_as_real(est)
    for ci, b in enumerate(blocks):
        d.update(sd(b, f"vn/w/cascades.{ci}."))
    # RSS variant of the final combine (same cascades)
    with torch.no_grad():
        est = y.clone()
        for b in blocks:
            est = b(est, y, S, m)
        est = utils.coil_combination(fft.ifft2(est, centered=False, normalization="backward"), S, method="RSS", dim=1)
    d["vn/out_rss_realview"] = est

    # UNet model: unet.py:108-121
    torch.manual_seed(801)
    ucfg = dict(channels=8, pooling_layers=2, padding_size=11, normalize=True, fft_centered=True,
                fft_normalization="ortho", spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE")
    nu = unet_block.NormUnet(8, 2, padding_size=11, normalize=True).eval()
    k2 = fft.fft2(utils.complex_mul(img, S), centered=True, normalization="ortho")
    y2 = k2 * m
    with torch.no_grad():
        eta = torch.view_as_complex(utils.coil_combination(fft.ifft2(y2, centered=True, normalization="ortho"), S,
                                                           method="SENSE", dim=1))
        _, eta = utils.center_crop_to_smallest(target, eta)
        out = torch.view_as_complex(nu(torch.view_as_real(eta.unsqueeze(1)))).squeeze(1)
    d["unet/cfg"] = np.array(json.dumps(ucfg))
    d["unet/y"] = y2
    d["unet/out"] = torch.view_as_real(out)
    d.update(sd(nu, "unet/w/unet."))

    # ZF: zf.py:90-100, both methods
    for meth in ("SENSE", "RSS"):
        pred = utils.coil_combination(fft.ifft2(y2, centered=True, normalization="ortho"), S, method=meth, dim=1)
        pred = utils.check_stacked_complex(pred)
        _, pred = utils.center_crop_to_smallest(target, pred)
        d[f"zf/out_{meth}"] = torch.view_as_real(pred) if pred.is_complex() else pred
    # C1: ZF + single soft-DC step on 1-coil 320x320 (checksums only)
    img1, S1 = synth(1, 1, 320, 320, 820)
    k1 = fft.fft2(utils.complex_mul(img1, S1), centered=False, normalization="backward")
    m1 = subsample.RandomMaskFunc([0.08], [4])([1, 320, 320, 2], seed=123)[0].reshape(1, 1, 1, 320, 1).bool()
    y1 = k1 * m1
    zf = utils.sense(fft.ifft2(y1, centered=False, normalization="backward"), S1, 1)
    kk = fft.fft2(utils.complex_mul(zf.unsqueeze(1), S1), centered=False, normalization="backward")
    dc = torch.where(m1, kk - y1, torch.zeros(1, 1, 1, 1, 1))
    d["c1/zf_sample"] = zf.reshape(-1)[::997].clone()
    d["c1/zf_l2"] = zf.double().norm().reshape(1)
    d["c1/dc_sample"] = dc.reshape(-1)[::997].clone()
    d["c1/dc_l2"] = dc.double().norm().reshape(1)
    save("g8_models.npz", d)


def g9_qrim():
    """qRIMBlock + analytical_log_likelihood_gradient + a 2-cascade qCIRIM composition (qcirim.py:248-312)."""
    qrim_block = _refshim.load("mridc.collections.quantitative.models.qrim.qrim_block")
    qutils = _refshim.load("mridc.collections.quantitative.models.qrim.utils")
    d = {}
    B, E, C, H, W = 2, 4, 3, 16, 12
    TEs = [3.0, 11.5, 20.0, 28.5]
    g = torch.Generator().manual_seed(900)
    r2 = torch.rand(B, H, W, generator=g) * 60 + 5
    s0 = torch.rand(B, H, W, generator=g) * 2 + 0.2
    b0 = (torch.rand(B, H, W, generator=g) - 0.5) * 80
    ph = (torch.rand(B, H, W, generator=g) - 0.5) * 1.0
    _, S = synth(B, C, H, W, 901)
    fm = qutils.SignalForwardModel(sequence="MEGRE")
    sig = fm(r2, s0, b0, ph, TEs)                                        # [B,E,H,W,2]
    k = fft.fft2(utils.complex_mul(sig.unsqueeze(2), S.unsqueeze(1)), centered=True, normalization="ortho")   # [B,E,C,H,W,2]
    _, m = make_mask([1, C, H, W, 2])
    m = torch.cat([m, torch.roll(m, 2, dims=-2)], 0).unsqueeze(1)          # [B,1,1,1,W,1]
    y = k * m + 0.01 * rnd([B, E, C, H, W, 2], 902) * m
    d.update(r2=r2, s0=s0, b0=b0, ph=ph, S=S, y=y, mask=m, TEs=np.array(TEs, dtype=np.float32), signal=sig)
    # perturbed initial maps -> non-trivial gradient
    r2i, s0i, b0i, phi_i = r2 * 0.8 + 3, s0 * 1.1, b0 * 0.9 + 2, ph * 0.7
    for cen, norm in ((True, "ortho"), (False, "backward")):
        kk = fft.fft2(utils.complex_mul(sig.unsqueeze(2), S.unsqueeze(1)), centered=cen, normalization=norm)
        yy = kk * m
        grads = torch.stack([qutils.analytical_log_likelihood_gradient(fm, r2i[i], s0i[i], b0i[i], phi_i[i], TEs, S[i], yy[i],
                                                                      m[i], cen, norm, [-2, -1], 2) for i in range(B)])
        d[f"grad/{int(cen)}_{norm}/y"] = yy
        d[f"grad/{int(cen)}_{norm}/out"] = grads
    d.update(r2i=r2i, s0i=s0i, b0i=b0i, phi_i=phi_i)
    gamma = [150.0, 150.0, 1000.0, 150.0]
    cfg = dict(quantitative_module_recurrent_layer="IndRNN", quantitative_module_conv_filters=[32, 32, 4],
               quantitative_module_conv_kernels=[5, 3, 3], quantitative_module_conv_dilations=[1, 2, 1],
               quantitative_module_conv_bias=[True, True, False], quantitative_module_recurrent_filters=[32, 32, 0],
               quantitative_module_recurrent_kernels=[1, 1, 0], quantitative_module_recurrent_dilations=[1, 1, 0],
               quantitative_module_recurrent_bias=[True, True, False], quantitative_module_depth=2,
               quantitative_module_time_steps=3, quantitative_module_num_cascades=2, quantitative_module_no_dc=True,
               quantitative_module_signal_forward_model_sequence="MEGRE", quantitative_module_dimensionality=2,
               quantitative_module_gamma_regularization_factors=gamma, use_reconstruction_module=False,
               fft_centered=True, fft_normalization="ortho", spatial_dims=[-2, -1], coil_dim=2, coil_combination_method="SENSE")
    torch.manual_seed(910)
    blocks = [qrim_block.qRIMBlock(
        recurrent_layer="IndRNN", conv_filters=[32, 32, 4], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1],
        conv_bias=[True, True, False], recurrent_filters=[32, 32, 0], recurrent_kernels=[1, 1, 0], recurrent_dilations=[1, 1, 0],
        recurrent_bias=[True, True, False], depth=2, time_steps=3, conv_dim=2, no_dc=True, linear_forward_model=fm,
        fft_centered=True, fft_normalization="ortho", spatial_dims=[-2, -1], coil_dim=2, coil_combination_method="SENSE",
        dimensionality=2).eval() for _ in range(2)]
    for b in blocks:
        scale_weights(b, 6.0)
    gm = torch.tensor(gamma)
    with torch.no_grad():
        # qcirim.py:248-312
        r2p, s0p, b0p, php = r2i / gm[0], s0i / gm[1], b0i / gm[2], phi_i / gm[3]
        eta, hx = None, None
        outs = []
        for i, cas in enumerate(blocks):
            prediction, hx = cas(y.clone(), y, r2p, s0p, b0p, php, TEs, S, m, eta, hx, gm, keep_eta=i != 0)
            r2p, s0p, b0p, php = (prediction[-1][:, 0], prediction[-1][:, 1], prediction[-1][:, 2], prediction[-1][:, 3])
            outs.append(torch.stack([qutils.RescaleByMax.reverse(torch.abs(pp), gm) for pp in prediction]))
            if i == 0:
                d["block0/etas"] = torch.stack(prediction)
    d["qcirim/cfg"] = np.array(json.dumps(cfg))
    d["qcirim/out"] = torch.stack(outs)                                   # [cascade, step, B, 4, H, W]
    for ci, b in enumerate(blocks):
        d.update(sd(b, f"qcirim/w/qcirim.{ci}."))
    save("g9_qrim.npz", d)


def g10_ssim():
    d = {}
    loss = ssim_mod.SSIMLoss()
    X = torch.rand(2, 1, 20, 17, generator=torch.Generator().manual_seed(1001))
    Y = (X + 0.1 * torch.rand(2, 1, 20, 17, generator=torch.Generator().manual_seed(1002))).clamp(0, 1)
    dr = torch.tensor([1.0, 0.8])
    d.update(X=X, Y=Y, data_range=dr)
    d["loss"] = loss(X, Y, dr).reshape(1)
    d["loss_same"] = loss(X, X, dr).reshape(1)
    save("g10_ssim.npz", d)


def _randomize_bn(mod, seed):
    """Non-trivial BatchNorm running statistics / affine (eval mode uses them)."""
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.randn(m.num_features, generator=g) * 0.3 + 1.0)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)


def g15_cascadenet():
    """N4: conv/conv2d.py:8-69, cascadenet/ccnn_block.py:10-139, ccnn.py:116-142 (model forward composed from the blocks)."""
    conv2d = _refshim.load("mridc.collections.reconstruction.models.conv.conv2d")
    ccnn_block = _refshim.load("mridc.collections.reconstruction.models.cascadenet.ccnn_block")
    d = {}
    cases = [("h16n3", 16, 3, False, [1, 3, 32, 16, 2], True, False), ("h8n4_bn", 8, 4, True, [1, 5, 15, 12, 2], False, False),
             ("h12n2_nodc", 12, 2, False, [2, 2, 21, 26, 2], False, True)]
    for i, (nm, hid, nconv, bn, shape, centered, no_dc) in enumerate(cases):
        torch.manual_seed(1500 + i)
        act = torch.nn.PReLU()
        with torch.no_grad():
            act.weight.fill_(0.25 - 0.05 * i)
        net = conv2d.Conv2d(2, 2, hid, n_convs=nconv, activation=act, batchnorm=bn)
        _randomize_bn(net, 1510 + i)
        norm = "ortho" if centered else "backward"
        blk = ccnn_block.CascadeNetBlock(net, fft_centered=centered, fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1,
                                         no_dc=no_dc).eval()
        with torch.no_grad():
            blk.dc_weight.fill_(0.8)
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 1520 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        pred = y + 0.1 * rnd(list(y.shape), 1530 + i)
        mm = m.byte() if i % 2 == 0 else m.bool()
        with torch.no_grad():
            x_in = rnd([B, 2, H, W], 1540 + i)
            d[f"{nm}/conv_in"], d[f"{nm}/conv_out"] = x_in, net(x_in)
            d[f"{nm}/conv_out_5d"] = net(x_in.permute(0, 2, 3, 1).unsqueeze(1).contiguous())     # conv2d.py:64-67
            out = blk(pred, y, S, mm)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(hidden_channels=hid, n_convs=nconv, batchnorm=bn, fft_centered=centered,
                                                  fft_normalization=norm, no_dc=no_dc)))
        d[f"{nm}/pred"], d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"], d[f"{nm}/out"] = pred, y, S, mm, out
        d.update(sd(blk, f"{nm}/w/"))
    # model: ccnn.py:116-142 with 3 cascades
    torch.manual_seed(1550)
    cfg = dict(num_cascades=3, hidden_channels=10, n_convs=3, batchnorm=False, no_dc=False, fft_centered=False,
               fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, coil_combination_method="SENSE")
    blocks = [ccnn_block.CascadeNetBlock(conv2d.Conv2d(2, 2, 10, n_convs=3, activation=torch.nn.PReLU(), batchnorm=False),
                                         fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1,
                                         no_dc=False).eval() for _ in range(3)]
    B, C, H, W = 1, 4, 30, 22
    img, S = synth(B, C, H, W, 1560)
    k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
    _, m = make_mask([1, C, H, W, 2])
    m = m.byte()
    y = k * m
    target = utils.center_crop(utils.complex_abs(utils.sense(fft.ifft2(k, centered=False, normalization="backward"), S, 1)), (26, 20))
    with torch.no_grad():
        pred = y.clone()
        for b in blocks:
            pred = b(pred, y, S, m)
        pred = torch.view_as_complex(utils.coil_combination(fft.ifft2(pred, centered=False, normalization="backward",
                                                                      spatial_dims=[-2, -1]), S, method="SENSE", dim=1))
        _, pred = utils.center_crop_to_smallest(target, pred)
    d["model/cfg"] = np.array(json.dumps(cfg))
    d["model/y"], d["model/S"], d["model/mask"], d["model/target"] = y, S, m, target
    d["model/out"] = torch.view_as_real(pred)
    for ci, b in enumerate(blocks):
        d.update(sd(b, f"model/w/cascades.{ci}."))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g15_cascadenet.npz", d)


def g16_vsnet():
    """N4: variablesplittingnet/vsnet_block.py:12-146 and vsnet.py:143-167 (CONV denoiser shared by all cascades, vsnet.py:81-83)."""
    conv2d = _refshim.load("mridc.collections.reconstruction.models.conv.conv2d")
    vsb = _refshim.load("mridc.collections.reconstruction.models.variablesplittingnet.vsnet_block")
    d = {}
    cases = [("c3", 3, 12, 3, [1, 3, 24, 16, 2], False), ("c2_centered", 2, 8, 2, [1, 4, 15, 18, 2], True)]
    for i, (nm, ncasc, hid, nconv, shape, centered) in enumerate(cases):
        torch.manual_seed(1600 + i)
        norm = "ortho" if centered else "backward"
        den = conv2d.Conv2d(2, 2, hid, n_convs=nconv, activation=torch.nn.PReLU(), batchnorm=False)
        dcl, wat = vsb.DataConsistencyLayer(), vsb.WeightedAverageTerm()
        with torch.no_grad():
            dcl.dc_weight.fill_(0.9)
            wat.param.fill_(0.6)
        blk = vsb.VSNetBlock(torch.nn.ModuleList([den] * ncasc), torch.nn.ModuleList([dcl] * ncasc),
                             torch.nn.ModuleList([wat] * ncasc), num_cascades=ncasc, fft_centered=centered, fft_normalization=norm,
                             spatial_dims=[-2, -1], coil_dim=1).eval()
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 1610 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=centered, normalization=norm), S, 1))
        with torch.no_grad():
            out = blk(y, S, m)
            d[f"{nm}/dc"] = dcl(out, y, m)
            d[f"{nm}/wa"] = wat(out, y)
            image = torch.view_as_complex(utils.coil_combination(fft.ifft2(out, centered=centered, normalization=norm,
                                                                           spatial_dims=[-2, -1]), S, method="SENSE", dim=1))
            _, image = utils.center_crop_to_smallest(target, image)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(num_cascades=ncasc, imspace_model_architecture="CONV", imspace_conv_hidden_channels=hid,
                                                  imspace_conv_n_convs=nconv, imspace_conv_batchnorm=False, fft_centered=centered,
                                                  fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1,
                                                  coil_combination_method="SENSE")))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"], d[f"{nm}/target"] = y, S, m, target
        d[f"{nm}/block_out"], d[f"{nm}/model_out"] = out, torch.view_as_real(image)
        d.update(sd(blk, f"{nm}/w/model."))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g16_vsnet.npz", d)


def g17_dc_layers():
    """N4: sigmanet/dc_layers.py:22-478 forward passes (batch 1: the layers index the coil axis as -4 and the batch axis as -5)."""
    dcl = _refshim.load("mridc.collections.reconstruction.models.sigmanet.dc_layers")
    d = {}
    cases = [("uncentered", False, "backward", [1, 3, 20, 16]), ("centered", True, "ortho", [1, 4, 15, 18])]
    for i, (nm, centered, norm, (B, C, H, W)) in enumerate(cases):
        img, S = synth(B, C, H, W, 1700 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        x = utils.sense(fft.ifft2(y, centered=centered, normalization=norm), S, 1) + 0.05 * rnd([B, H, W, 2], 1710 + i)
        kw = dict(fft_centered=centered, fft_normalization=norm, spatial_dims=[-2, -1])
        with torch.no_grad():
            d[f"{nm}/gd"] = dcl.DataGDLayer(0.3, **kw)(x, y, S, m)
            d[f"{nm}/vs"] = dcl.DataVSLayer(0.4, 0.7, **kw)(x, y, S, m)
            d[f"{nm}/dc_single"] = dcl.DCLayer(0.2, **kw)(x, y[:, 0], m[:, 0])
            for it in (3, 10):
                # the CG layer only runs with a 5-D image [B,1,H,W,2] (its solver reshapes alpha to 5-D, dc_layers.py:191-192)
                d[f"{nm}/prox{it}"] = dcl.DataProxCGLayer(0.5, tol=1e-6, iter=it, **kw)(x.unsqueeze(1), y, S, m)
        d[f"{nm}/x"], d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = x, y, S, m
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(fft_centered=centered, fft_normalization=norm)))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g17_dc_layers.npz", d)


def g21_dunet():
    """N4: didn/didn.py:10-325, sigmanet/sensitivity_net.py:17-212 and the model forward dunet.py:162-176, composed from the imported
    blocks as DUNet.__init__ does (dunet.py:47-107: ONE DIDN and ONE data layer, wrapped / listed num_iter times).  Batch 1 and the
    proximal-CG data term, as in the reference's own test: the other data layers do not accept the 5-D image the wrapper produces."""
    didn = _refshim.load("mridc.collections.reconstruction.models.didn.didn")
    snet = _refshim.load("mridc.collections.reconstruction.models.sigmanet.sensitivity_net")
    dcl = _refshim.load("mridc.collections.reconstruction.models.sigmanet.dc_layers")
    d = {}
    # the regulariser on its own: odd sizes (reflect padding inside the down-up blocks), batch 2, with and without the skip connection
    for i, (nm, hid, dubs, convs, shape, skip) in enumerate([("didn_a", 8, 2, 3, [2, 2, 17, 19], False), ("didn_b", 16, 1, 1, [1, 2, 24, 32], True)]):
        torch.manual_seed(2100 + i)
        net = didn.DIDN(2, 2, hidden_channels=hid, num_dubs=dubs, num_convs_recon=convs, skip_connection=skip).eval()
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.PReLU):
                    m.weight.uniform_(0.05, 0.45)
        x = rnd(shape, 2110 + i)
        with torch.no_grad():
            d[f"{nm}/out"] = net(x)
        d[f"{nm}/x"] = x
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(hidden_channels=hid, num_dubs=dubs, num_convs_recon=convs, skip_connection=skip)))
        d.update(sd(net, f"{nm}/w/"))
    cases = [("prox_unshared", [1, 3, 32, 16], dict(num_iter=2, didn_hidden_channels=8, didn_num_dubs=1, didn_num_convs_recon=1,
                                                    data_consistency_iterations=4, shared_params=False), True, "ortho"),
             ("prox_shared", [1, 2, 17, 19], dict(num_iter=3, didn_hidden_channels=8, didn_num_dubs=2, didn_num_convs_recon=2,
                                                  data_consistency_iterations=6, shared_params=True), False, "backward")]
    for i, (nm, (B, C, H, W), c, centered, norm) in enumerate(cases):
        torch.manual_seed(2150 + i)
        cfg = dict(c, reg_model_architecture="DIDN", data_consistency_term="PROX", data_consistency_lambda_init=0.1, fft_centered=centered,
                   fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1, use_sens_net=False, coil_combination_method="SENSE",
                   train_loss_fn="l1", val_loss_fn="l1")
        reg = didn.DIDN(2, 2, hidden_channels=cfg["didn_hidden_channels"], num_dubs=cfg["didn_num_dubs"],
                        num_convs_recon=cfg["didn_num_convs_recon"])
        dc = dcl.DataProxCGLayer(lambda_init=0.1, iter=cfg["data_consistency_iterations"], fft_centered=centered, fft_normalization=norm,
                                 spatial_dims=[-2, -1])
        net = snet.SensitivityNetwork(cfg["num_iter"], reg, dc, shared_params=cfg["shared_params"], save_space=False, reset_cache=False).eval()
        img, S = synth(B, C, H, W, 2160 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=centered, normalization=norm), S, 1))
        with torch.no_grad():
            init_pred = torch.sum(utils.complex_mul(fft.ifft2(y, centered=centered, normalization=norm, spatial_dims=[-2, -1]),
                                                    utils.complex_conj(S)), 1)
            x_net = net(init_pred, y, S, m)
            image = torch.sum(utils.complex_mul(x_net, utils.complex_conj(S)), 1)
            image = torch.view_as_complex(image)
            _, image = utils.center_crop_to_smallest(target, image)
            d[f"{nm}/reg0"] = net.gradR[0](init_pred)                      # the normalisation wrapper on the 4-D first iterate
        d[f"{nm}/cfg"] = np.array(json.dumps(cfg))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"], d[f"{nm}/target"] = y, S, m, target
        d[f"{nm}/init_pred"], d[f"{nm}/net_out"], d[f"{nm}/model_out"] = init_pred, x_net, torch.view_as_real(image)
        d.update(sd(net, f"{nm}/w/model."))
    d["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g21_dunet.npz", d)


def g18_rvn():
    """N4: recurrentvarnet/conv2gru.py:10-163, recurrentvarnet/recurrentvarnet.py:17-240, rvn.py:163-226 (model forward composed
    from the imported blocks; hidden size 16 keeps the fixture small)."""
    c2g = _refshim.load("mridc.collections.reconstruction.models.recurrentvarnet.conv2gru")
    rv = _refshim.load("mridc.collections.reconstruction.models.recurrentvarnet.recurrentvarnet")
    d = {}
    # Conv2dGRU alone: zero state (None) and a given state; gru kernel 1 (the block's) and 3
    for i, (nm, cin, hid, nl, gk, shape) in enumerate([("gru_h16_l2", 2, 16, 2, 1, [2, 13, 18]), ("gru_h8_l4", 2, 8, 4, 1, [1, 20, 16]),
                                                      ("gru_h8_l2_k3", 3, 8, 2, 3, [1, 9, 11])]):
        torch.manual_seed(1800 + i)
        net = c2g.Conv2dGRU(cin, hid, num_layers=nl, gru_kernel_size=gk, replication_padding=True).eval()
        B, H, W = shape
        x = rnd([B, cin, H, W], 1810 + i)
        st = rnd([B, hid, H, W, nl], 1820 + i, 0.5)
        with torch.no_grad():
            o0, s0 = net(x, None)
            o1, s1 = net(x, st)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(in_channels=cin, hidden_channels=hid, num_layers=nl, gru_kernel_size=gk)))
        d[f"{nm}/x"], d[f"{nm}/state"] = x, st
        d[f"{nm}/out0"], d[f"{nm}/state0"], d[f"{nm}/out1"], d[f"{nm}/state1"] = o0, s0, o1, s1
        d.update(sd(net, f"{nm}/w/"))
    # RecurrentInit
    for i, (nm, chans, dils, depth, ms) in enumerate([("init_ms1", (8, 8, 16, 16), (1, 1, 2, 4), 4, 1), ("init_ms3", (4, 6, 8), (1, 2, 1), 2, 3)]):
        torch.manual_seed(1830 + i)
        ini = rv.RecurrentInit(2, 12, channels=chans, dilations=dils, depth=depth, multiscale_depth=ms).eval()
        x = rnd([1, 2, 17, 14], 1840 + i)
        with torch.no_grad():
            out = ini(x)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(in_channels=2, out_channels=12, channels=list(chans), dilations=list(dils), depth=depth,
                                                  multiscale_depth=ms)))
        d[f"{nm}/x"], d[f"{nm}/out"] = x, out
        d.update(sd(ini, f"{nm}/w/"))
    # block + model (rvn.py:163-226): learned initializer "sense", 8 steps, shared and unshared parameters
    for i, (nm, share, centered, (B, C, H, W)) in enumerate([("model_shared", True, False, (1, 3, 24, 16)),
                                                             ("model_unshared", False, True, (1, 4, 15, 18))]):
        torch.manual_seed(1850 + i)
        norm = "ortho" if centered else "backward"
        hid, nl, steps = 12, 3, 8
        ini = rv.RecurrentInit(2, hid, channels=(6, 6, 8, 8), dilations=(1, 1, 2, 4), depth=nl, multiscale_depth=1).eval()
        blocks = [rv.RecurrentVarNetBlock(in_channels=2, hidden_channels=hid, num_layers=nl, fft_centered=centered,
                                          fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1).eval()
                  for _ in range(1 if share else steps)]
        for b_ in blocks:
            with torch.no_grad():
                b_.learning_rate.fill_(0.8)
        img, S = synth(B, C, H, W, 1860 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=centered, normalization=norm)
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=centered, normalization=norm), S, 1))
        with torch.no_grad():
            init_img = utils.complex_mul(fft.ifft2(y, centered=centered, normalization=norm, spatial_dims=[-2, -1]),
                                         utils.complex_conj(S)).sum(1).unsqueeze(1)
            state = ini(fft.fft2(init_img, centered=centered, normalization=norm, spatial_dims=[-2, -1]).sum(1).permute(0, 3, 1, 2))
            d[f"{nm}/init_state"] = state
            kp = y.clone()
            for step in range(steps):
                kp, state = blocks[0 if share else step](kp, y, m, S, state)
                if step == 0:
                    d[f"{nm}/k_step0"], d[f"{nm}/state_step0"] = kp, state
            eta = fft.ifft2(kp, centered=centered, normalization=norm, spatial_dims=[-2, -1])
            eta = torch.view_as_complex(utils.coil_combination(eta, S, method="SENSE", dim=1))
            _, eta = utils.center_crop_to_smallest(target, eta)
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(in_channels=2, recurrent_hidden_channels=hid, recurrent_num_layers=nl, num_steps=steps,
                                                  no_parameter_sharing=not share, learned_initializer=True,
                                                  initializer_initialization="sense", initializer_channels=[6, 6, 8, 8],
                                                  initializer_dilations=[1, 1, 2, 4], initializer_multiscale=1, fft_centered=centered,
                                                  fft_normalization=norm, spatial_dims=[-2, -1], coil_dim=1,
                                                  coil_combination_method="SENSE", pretrained=True)))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"], d[f"{nm}/target"] = y, S, m, target
        d[f"{nm}/k_final"], d[f"{nm}/out"] = kp, torch.view_as_real(eta)
        d.update(sd(ini, f"{nm}/w/initializer."))
        for bi, b_ in enumerate(blocks):
            d.update(sd(b_, f"{nm}/w/block_list.{bi}."))
    d["names"] = np.array(json.dumps(["gru_h16_l2", "gru_h8_l4", "gru_h8_l2_k3", "init_ms1", "init_ms3", "model_shared", "model_unshared"]))
    save("g18_rvn.npz", d)


def g22_precision16():
    """The reference's OWN blocks under `torch.autocast("cpu", dtype=torch.float16)` -- what pytorch-lightning's native AMP wraps the forward pass in for
    `trainer.precision: 16` (projects/reconstruction/model_zoo/conf/base_cirim_run.yaml:132, base_vn_run.yaml:98): a RIMBlock (the model-zoo shape: IndRNN, 64
    filters; one case with recurrent weights x 5) and a VarNetBlock with its NormUnet.  Pins `oracle.amp.autocast_fp16` (the checker of the HIP precision-16
    routes) to the reference itself; outputs stored as float32."""
    d = {}
    names = []
    for i, (nm, ov, shape, wscale) in enumerate([("rim_ind64", dict(time_steps=4), [1, 4, 24, 20, 2], 1.0),
                                                  ("rim_ind64_x5", dict(time_steps=8, fft_centered=False, fft_normalization="backward"), [1, 3, 16, 12, 2], 5.0)]):
        cfg = dict(RIM_CFG)
        cfg.update(ov)
        torch.manual_seed(2200 + i)
        blk = rim_block.RIMBlock(**cfg).eval()
        scale_weights(blk, wscale)
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 2210 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=cfg["fft_centered"], normalization=cfg["fft_normalization"])
        _, m = make_mask([1, C, H, W, 2])
        m = m.bool()
        y = k * m
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            outs, hx = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
        with torch.no_grad():
            outs32, _ = blk(y, y, S, m, None, None, 1.0, keep_eta=False)
        d[f"{nm}/cfg"] = np.array(json.dumps(cfg))
        d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = y, S, m
        d[f"{nm}/outs"] = torch.stack([o.float() for o in outs])
        d[f"{nm}/outs_fp32"] = torch.stack(outs32)
        for j, h in enumerate(hx):
            d[f"{nm}/hx{j}"] = h.float()
        d.update(sd(blk, f"{nm}/w/"))
        names.append(nm)
    for i, (nm, ch, pools, pad, shape) in enumerate([("vn_u14p2", 14, 2, 11, [1, 3, 32, 16, 2]), ("vn_u8p3_odd", 8, 3, 7, [1, 4, 21, 26, 2])]):
        torch.manual_seed(2250 + i)
        nu = unet_block.NormUnet(ch, pools, padding_size=pad, normalize=True)
        blk = vn_block.VarNetBlock(nu, fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=1, no_dc=False).eval()
        B, C, H, W, _ = shape
        img, S = synth(B, C, H, W, 2260 + i)
        k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
        _, m = make_mask([1, C, H, W, 2])
        y = k * m
        pred = y + 0.1 * rnd(list(y.shape), 2270 + i)
        with torch.no_grad():
            eta_in = blk.sens_reduce(pred, S)
            with torch.autocast("cpu", dtype=torch.float16):
                nu_out = nu(eta_in)
                out = blk(pred, y, S, m.bool())
            out32 = blk(pred, y, S, m.bool())
        d[f"{nm}/cfg"] = np.array(json.dumps(dict(chans=ch, num_pools=pools, padding_size=pad, normalize=True, fft_centered=False,
                                                  fft_normalization="backward", no_dc=False)))
        d[f"{nm}/pred"], d[f"{nm}/y"], d[f"{nm}/S"], d[f"{nm}/mask"] = pred, y, S, m.bool()
        d[f"{nm}/eta_in"], d[f"{nm}/normunet_out"], d[f"{nm}/out"], d[f"{nm}/out_fp32"] = eta_in, nu_out.float(), out.float(), out32
        d.update(sd(blk, f"{nm}/w/"))
        names.append(nm)
    # a one-cascade qCIRIM (base_qcirim_run.yaml:204 `precision: 16`) with the model-zoo widths (IndRNN, 128 filters): qcirim.py:248-312 around the reference's qRIMBlock
    qrim_block = _refshim.load("mridc.collections.quantitative.models.qrim.qrim_block")
    qutils = _refshim.load("mridc.collections.quantitative.models.qrim.utils")
    B, E, C, H, W = 1, 4, 4, 24, 20
    TEs = [3.0, 11.5, 20.0, 28.5]
    g = torch.Generator().manual_seed(2290)
    r2 = torch.rand(B, H, W, generator=g) * 60 + 5
    s0 = torch.rand(B, H, W, generator=g) * 2 + 0.2
    b0 = (torch.rand(B, H, W, generator=g) - 0.5) * 80
    ph = (torch.rand(B, H, W, generator=g) - 0.5) * 1.0
    _, S = synth(B, C, H, W, 2291)
    fm = qutils.SignalForwardModel(sequence="MEGRE")
    sig = fm(r2, s0, b0, ph, TEs)
    k = fft.fft2(utils.complex_mul(sig.unsqueeze(2), S.unsqueeze(1)), centered=False, normalization="backward")
    _, m = make_mask([1, C, H, W, 2])
    m = m.unsqueeze(1)
    y = k * m + 0.01 * rnd([B, E, C, H, W, 2], 2292) * m
    r2i, s0i, b0i, phi_i = r2 * 0.8 + 3, s0 * 1.1, b0 * 0.9 + 2, ph * 0.7
    gamma = [150.0, 150.0, 1000.0, 150.0]
    qcfg = dict(quantitative_module_recurrent_layer="IndRNN", quantitative_module_conv_filters=[128, 128, 4],
                quantitative_module_conv_kernels=[5, 3, 3], quantitative_module_conv_dilations=[1, 2, 1],
                quantitative_module_conv_bias=[True, True, False], quantitative_module_recurrent_filters=[128, 128, 0],
                quantitative_module_recurrent_kernels=[1, 1, 0], quantitative_module_recurrent_dilations=[1, 1, 0],
                quantitative_module_recurrent_bias=[True, True, False], quantitative_module_depth=2,
                quantitative_module_time_steps=8, quantitative_module_num_cascades=1, quantitative_module_no_dc=True,
                quantitative_module_signal_forward_model_sequence="MEGRE", quantitative_module_dimensionality=2,
                quantitative_module_gamma_regularization_factors=gamma, use_reconstruction_module=False,
                fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=2, coil_combination_method="SENSE")
    torch.manual_seed(2293)
    qb = qrim_block.qRIMBlock(
        recurrent_layer="IndRNN", conv_filters=[128, 128, 4], conv_kernels=[5, 3, 3], conv_dilations=[1, 2, 1],
        conv_bias=[True, True, False], recurrent_filters=[128, 128, 0], recurrent_kernels=[1, 1, 0], recurrent_dilations=[1, 1, 0],
        recurrent_bias=[True, True, False], depth=2, time_steps=8, conv_dim=2, no_dc=True, linear_forward_model=fm,
        fft_centered=False, fft_normalization="backward", spatial_dims=[-2, -1], coil_dim=2, coil_combination_method="SENSE",
        dimensionality=2).eval()
    scale_weights(qb, 4.0)
    gm = torch.tensor(gamma)

    def run():
        prediction, _ = qb(y.clone(), y, r2i / gm[0], s0i / gm[1], b0i / gm[2], phi_i / gm[3], TEs, S, m, None, None, gm, keep_eta=False)
        return torch.stack([qutils.RescaleByMax.reverse(torch.abs(pp.float()), gm) for pp in prediction])
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.float16):
            out16 = run()
        out32 = run()
    d["qcirim/cfg"] = np.array(json.dumps(qcfg))
    d.update({"qcirim/r2i": r2i, "qcirim/s0i": s0i, "qcirim/b0i": b0i, "qcirim/phi_i": phi_i, "qcirim/y": y, "qcirim/S": S, "qcirim/mask": m,
              "qcirim/TEs": np.array(TEs, dtype=np.float32), "qcirim/out": out16.float(), "qcirim/out_fp32": out32})     # out: [step, B, 4, H, W]
    d.update(sd(qb, "qcirim/w/qcirim.0."))
    d["names"] = np.array(json.dumps(names))
    save("g22_precision16.npz", d)


def g23_training_autocast():
    """BASELINE config 4 (CIRIM training under `precision: 16` AMP, base_cirim_train.yaml:180 -- bf16 is the half type torch's CPU autocast trains in): the
    reference's own RIMBlocks composed as cirim.py:146-165, the loss of cirim.py:199-247 (l1, accumulate_estimates) and torch autograd through the reference
    modules, under `torch.autocast("cpu", dtype=torch.bfloat16)` and in fp32: loss and every parameter gradient.  Pins `oracle.amp.cirim_loss_and_gradients(...,
    mode="autocast_bf16")` -- one of the three arithmetics the HIP training tape is checked against -- to the reference."""
    import math
    d = {}
    cfg = dict(RIM_CFG)
    cfg.update(coil_combination_method="SENSE", keep_eta=True, num_cascades=2, time_steps=8, fft_centered=False, fft_normalization="backward", no_dc=True)
    T = 8 * math.ceil(cfg["time_steps"] / 8)
    bcfg = {k: v for k, v in cfg.items() if k not in ("coil_combination_method", "keep_eta", "num_cascades")}
    bcfg["time_steps"] = T
    torch.manual_seed(2300)
    blocks = [rim_block.RIMBlock(**bcfg) for _ in range(cfg["num_cascades"])]
    for b in blocks:
        scale_weights(b, 3.0)
    B, C, H, W = 1, 4, 24, 20
    img, S = synth(B, C, H, W, 2310)
    k = fft.fft2(utils.complex_mul(img, S), centered=False, normalization="backward")
    _, m = make_mask([1, C, H, W, 2])
    m = m.bool()
    y = k * m
    target = utils.complex_abs(utils.sense(fft.ifft2(k, centered=False, normalization="backward"), S, 1))
    l1 = torch.nn.L1Loss()

    def loss_and_grads(ctx):
        for b in blocks:
            b.zero_grad(set_to_none=True)
        with ctx:
            out = compose_cirim(blocks, cfg, y, S, m, None, target)
            tgt = torch.abs(target / torch.max(torch.abs(target)))
            closs = []
            for cp in out:
                ls = [l1(tgt, torch.abs(t / torch.max(torch.abs(t)))) for t in cp]
                _l = [x * torch.logspace(-1, 0, steps=T).to(ls[0]) for x in ls]
                closs.append(sum(sum(_l) / T))
            loss = sum(closs) / len(blocks)
        loss.backward()
        return loss.detach().float().reshape(1), {f"cirim.{ci}.{n}": p.grad.detach().float().clone() for ci, b in enumerate(blocks) for n, p in b.named_parameters()
                                                   if p.grad is not None}
    import contextlib
    loss16, g16 = loss_and_grads(torch.autocast("cpu", dtype=torch.bfloat16))
    loss32, g32 = loss_and_grads(contextlib.nullcontext())
    d["cfg"] = np.array(json.dumps(cfg))
    d["y"], d["S"], d["mask"], d["target"] = y, S, m, target
    d["loss_autocast_bf16"], d["loss_fp32"] = loss16, loss32
    for kname, v in g16.items():
        d["grad_autocast_bf16/" + kname] = v
    for kname, v in g32.items():
        d["grad_fp32/" + kname] = v
    for ci, b in enumerate(blocks):
        d.update(sd(b, f"w/cirim.{ci}."))
    save("g23_training_autocast.npz", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g11", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22", "g23"]
    fns = dict(g23=g23_training_autocast, g22=g22_precision16, g21=g21_dunet, g20=g20_rim3d, g19=g19_cirim_spec, g18=g18_rvn, g15=g15_cascadenet, g16=g16_vsnet, g17=g17_dc_layers, g12=g12_mask_generators, g13=g13_transforms, g14=g14_sensnet, g9=g9_qrim, g1=g1_fft, g2=g2_shift, g3=g3_complex, g11=g11_masks, g4=g4_llg, g5=g5_rimblock, g6=g6_cirim,
               g7=g7_varnet, g8=g8_models, g10=g10_ssim)
    for w in which:
        fns[w]()
