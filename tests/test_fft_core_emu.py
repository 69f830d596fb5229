"""CPU: the Stockham plan / stage index math shared with the HIP kernels (mridc_amd/csrc/fft_core.h), emulated on
the host with plain loops and checked against numpy float64.  Covers the sizes the reference tests use
(3,4,6,8,10,12,13,15,16,17,18,19,32), the headline sizes (640, 372, 320, 256) and awkward ones (primes, prime squares)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = tmp_path_factory.mktemp("emu") / "libfftemu.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", str(out),
                           os.path.join(HERE, "emu", "fft_emu.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.emu_fft.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_long]
    lib.emu_plan.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.emu_fft_ct.argtypes = lib.emu_fft.argtypes
    return lib


SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 18, 19, 24, 25, 31, 32, 49, 64, 77, 100, 121, 127, 256,
         320, 372, 640, 1024, 2 * 3 * 5 * 7 * 11]


@pytest.mark.parametrize("n", SIZES)
def test_plan_product(emu, n):
    r = (ctypes.c_int * 16)()
    ns = emu.emu_plan(n, r)
    assert ns >= 0
    assert int(np.prod([r[i] for i in range(ns)], dtype=np.int64)) == n


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("inverse", [0, 1])
def test_emulated_fft_matches_numpy(emu, n, inverse):
    rng = np.random.default_rng(n * 2 + inverse)
    nseq = 3
    x = (rng.standard_normal((nseq, n)) + 1j * rng.standard_normal((nseq, n))).astype(np.complex64)
    # row layout: sequence-major, stride 1
    buf = np.ascontiguousarray(x.copy())
    assert emu.emu_fft(buf.ctypes.data, n, nseq, n, 1, inverse, buf.size) == 0
    ref = (np.fft.ifft(x.astype(np.complex128), axis=1) * n) if inverse else np.fft.fft(x.astype(np.complex128), axis=1)
    assert np.linalg.norm(buf - ref) <= 3e-6 * np.linalg.norm(ref)
    # column layout: batch-fastest, element stride nseq
    buf = np.ascontiguousarray(x.T.copy())
    assert emu.emu_fft(buf.ctypes.data, n, nseq, 1, nseq, inverse, buf.size) == 0
    assert np.linalg.norm(buf.T - ref) <= 3e-6 * np.linalg.norm(ref)


@pytest.mark.parametrize("n", [372, 640, 320, 256, 512, 384, 368, 77, 30, 16, 13])
@pytest.mark.parametrize("inverse", [0, 1])
def test_compile_time_plans_match_numpy(emu, n, inverse):
    """fft_ct.h: the compile-time plans the HIP kernels instantiate (radix 8, in-register prime butterflies)."""
    rng = np.random.default_rng(7 * n + inverse)
    nseq = 3
    x = (rng.standard_normal((nseq, n)) + 1j * rng.standard_normal((nseq, n))).astype(np.complex64)
    ref = (np.fft.ifft(x.astype(np.complex128), axis=1) * n) if inverse else np.fft.fft(x.astype(np.complex128), axis=1)
    buf = np.ascontiguousarray(x.copy())
    assert emu.emu_fft_ct(buf.ctypes.data, n, nseq, n, 1, inverse, buf.size) == 0
    assert np.linalg.norm(buf - ref) <= 3e-6 * np.linalg.norm(ref)
    buf = np.ascontiguousarray(x.T.copy())
    assert emu.emu_fft_ct(buf.ctypes.data, n, nseq, 1, nseq, inverse, buf.size) == 0
    assert np.linalg.norm(buf.T - ref) <= 3e-6 * np.linalg.norm(ref)


# ---- pfa372.h: the prime-factor 12 x 31 form of the 372-point row transform and the gradient pipeline built on it -------------------
@pytest.fixture(scope="module")
def pfa(emu):
    emu.emu_pfa372_fft.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    emu.emu_pfa372_task.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    return emu


@pytest.mark.parametrize("inverse", [0, 1])
def test_pfa372_transform_matches_numpy(pfa, inverse):
    rng = np.random.default_rng(372 + inverse)
    x = (rng.standard_normal(372) + 1j * rng.standard_normal(372)).astype(np.complex64)
    out = np.zeros(372, np.complex64)
    pfa.emu_pfa372_fft(x.ctypes.data, out.ctypes.data, inverse)
    ref = np.fft.ifft(x.astype(np.complex128)) * 372 if inverse else np.fft.fft(x.astype(np.complex128))
    assert np.linalg.norm(out - ref) <= 2e-6 * np.linalg.norm(ref)


@pytest.mark.parametrize("Cg", [5, 3, 1])
@pytest.mark.parametrize("centered", [0, 1])
def test_pfa372_gradient_task_matches_numpy(pfa, Cg, centered):
    """One task (<= 5 coils of one row) of the kernel's pipeline, lane by lane on the host, against the definition
    sum_c conj(S_c) IFFT_W(m (FFT_W(eta S_c) - yt_c)) (rim_utils.py:44-62 with the H transforms cancelled) in float64."""
    rng = np.random.default_rng(10 * Cg + centered)
    W = 372
    c64 = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)  # noqa: E731
    eta, S, yt = c64(W), c64(Cg, W), c64(Cg, W) * 5
    mask = (rng.uniform(size=W) < 0.3).astype(np.float32)
    half = W // 2 if centered else 0
    scale_f, scale_i = 1.0 / np.sqrt(W), 1.0 / np.sqrt(W)
    out = np.zeros(W, np.complex64)
    pfa.emu_pfa372_task(eta.ctypes.data, S.ctypes.data, yt.ctypes.data, mask.ctypes.data, out.ctypes.data, Cg, half,
                        ctypes.c_float(scale_f), ctypes.c_float(scale_i))
    sh = (lambda a: np.fft.ifftshift(a, axes=-1)) if centered else (lambda a: a)
    ush = (lambda a: np.fft.fftshift(a, axes=-1)) if centered else (lambda a: a)
    e, s_, y_ = eta.astype(np.complex128), S.astype(np.complex128), yt.astype(np.complex128)
    k = ush(np.fft.fft(sh(e[None] * s_), axis=-1)) * scale_f
    r = ush(np.fft.ifft(sh(mask[None] * (k - y_)), axis=-1)) * W * scale_i
    ref = (r * np.conj(s_)).sum(0)
    assert np.linalg.norm(out - ref) <= 3e-6 * np.linalg.norm(ref)
