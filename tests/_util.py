"""Shared helpers for the parity tests (tolerances follow SURVEY.md appendix C)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated tolerances (fp32 path): per-operator rel-L2 <= 1e-5 and |err| <= 1e-5 * max|ref| + rtol 1e-4;
# full 64-step fp32 CIRIM chain rel-L2 <= 1e-4.  Index/mask/select work: bit-exact.
OP_REL_L2 = 1e-5
CHAIN_REL_L2 = 1e-4


class Golden:
    def __init__(self):
        self._cache = {}

    def __call__(self, name):
        if name not in self._cache:
            self._cache[name] = np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
        return self._cache[name]


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def meta(z, key):
    return json.loads(str(z[key]))


def weights(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


def rel_l2(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def assert_close(got, ref, rel=OP_REL_L2, what=""):
    """Norm-relative criterion (elementwise rtol is wrong for FFT outputs)."""
    g = got.detach().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(got)
    r = ref.detach().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(ref)
    if g.is_complex():
        g = torch.view_as_real(g)
    if r.is_complex():
        r = torch.view_as_real(r)
    assert tuple(g.shape) == tuple(r.shape), f"{what}: shape {tuple(g.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(g).all(), f"{what}: non-finite values"
    e = rel_l2(g, r)
    assert e <= rel, f"{what}: rel-L2 {e:.3e} > {rel:.1e}"
    peak = float(r.abs().max()) if r.numel() else 0.0
    mx = float((g.double() - r.double()).abs().max()) if r.numel() else 0.0
    assert mx <= 1e-4 * peak * max(1.0, rel / OP_REL_L2) + 1e-30, f"{what}: max-abs {mx:.3e} vs peak {peak:.3e}"
    return e


def assert_exact(got, ref, what=""):
    g = got.detach().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(got)
    r = ref.detach().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(ref)
    assert tuple(g.shape) == tuple(r.shape), f"{what}: shape {tuple(g.shape)} vs {tuple(r.shape)}"
    r = r.to(g.dtype)
    if not torch.equal(g, r):
        bad = (g != r)
        diff = (g.double() - r.double()).abs()
        raise AssertionError(f"{what}: not bit-exact: {int(bad.sum())} of {g.numel()} elements differ, max |diff| "
                             f"{float(diff.max()):.3e} (max |ref| {float(r.double().abs().max()):.3e})")


def detie_terms(t, vals, margin, lo=None, hi=None):
    """Copy of the real tensor `t` in which no element is within `margin` of the same element of any tensor stacked in `vals` [E, *t.shape]
    (only the tied elements move, in steps of `margin`, staying inside (lo, hi))."""
    t = t.detach().clone()
    bad = ((vals - t.unsqueeze(0)).abs() < margin).any(0)
    for idx in bad.nonzero().tolist():
        idx = tuple(idx)
        col, cur = vals[(slice(None),) + idx], float(t[idx])
        for k in range(1, 400):
            cands = [c for c in (cur - k * margin, cur + k * margin) if (lo is None or c > lo) and (hi is None or c < hi)
                     and float((col - c).abs().min()) >= margin]
            if cands:
                t[idx] = cands[0]
                break
        else:
            raise AssertionError("no tie-free value for an element of the target")
    assert float((vals - t.unsqueeze(0)).abs().min()) >= 0.5 * margin
    return t


def detie_l1_target(target, preds, margin=1e-3):
    """Gradient tests of the reference's l1 loss (cirim.py:218-237: mean | t / max t - |p| / max |p| |) compare two fp32 implementations whose
    forward results differ by ~1e-7; the derivative of a term is its SIGN, so a term within round-off of zero -- or two pixels tying for max |p| --
    makes the comparison a coin toss that says nothing about the kernels.  Returns a copy of `target` (real, >= 0, [B,H,W]) in
    which every term of every estimate in `preds` (list of lists of complex [B,H,W], the oracle's forward) is at least `margin` away from zero,
    moving only the tied pixels and never the maximum; asserts that every estimate's largest modulus leads the runner-up by 1e-4 (relative)."""
    t = target.detach().clone().float()
    tmax = float(t.abs().max())
    pn = []
    for cascade in preds:
        for p in cascade:
            a = p.detach().abs().float()
            top = torch.topk(a.reshape(-1), 2).values
            assert float(top[0] - top[1]) > 1e-4 * float(top[0]), "two pixels tie for max |p|: pick another slice for this test"
            pn.append(a / top[0])
    out = detie_terms((t / tmax).abs(), torch.stack(pn), margin, lo=0.0, hi=1.0 - margin) * tmax
    out = torch.where(out == out, out, t)
    keep = (t / tmax).abs() >= 1.0 - margin                 # the maximum (and anything that close to it) stays exactly what it was
    out = torch.where(keep, t, out)
    assert float(out.abs().max()) == tmax
    return out
