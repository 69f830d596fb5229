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
