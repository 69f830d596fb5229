"""Build libmridc_amd.so (HIP, gfx950) in-tree with hipcc.  Called by __graft_entry__.build() and `python -m mridc_amd._build`."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmridc_amd.so")

# NO packed-fp32 vector instructions (v_pk_add / mul / fma_f32) in the FFT kernels (and the pointwise sources, where dropping them is bit-neutral): measured on MI355X (tools/probe/mfma_pk_interference.py), these
# kernels return WRONG results while a wave of another kernel on the same SIMD issues XDL MFMAs (two streams: the transforms of one slice next to
# the U-Net / few-channel convolutions of another -- 1e-3 .. 5e-2 errors, bit-exact when either side is alone) and are bit-exact in every
# combination once built without them.  The target feature stops the compiler from forming them, MRX_NO_PACKED_FP32 selects the scalar complex
# layer of pfa372.h instead of its inline assembly.  (Kernels of the other sources in which the compiler forms a few packed operations --
# transposed convolution, pooling, normalisation -- were tested as victims and are not affected; tests/test_gpu_concurrent_streams.py.)
NO_PACKED_FP32 = ["-DMRX_NO_PACKED_FP32", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# (source, extra flags).  elementwise.hip is built without fp contraction so that the pointwise complex
# operators round exactly like the reference's separate torch ops (mul, mul, sub).
SOURCES = [
    ("api.cpp", []),
    ("host_masks.cpp", []),
    ("fft.hip", NO_PACKED_FP32),
    ("llg372.hip", ["-fno-slp-vectorize"] + NO_PACKED_FP32),
    ("elementwise.hip", ["-ffp-contract=off"] + NO_PACKED_FP32),     # (bit-neutral here: without contraction a packed op is two scalar ones)
    ("conv.hip", []),
    ("rim_layer.hip", []),
    ("rim_layer_wino.hip", []),
    ("rim_layer1_sb.hip", []),
    ("rim_layer2_sb.hip", []),
    ("gated_cell.hip", []),
    ("gated_cell_sb.hip", []),
    ("conv_bwd.hip", []),
    ("conv_bf16.hip", []),
    ("conv_sbs.hip", []),
    ("unet.hip", []),
    ("unet_fused.hip", []),
    ("unet_f16.hip", []),
    ("qmri.hip", ["-ffp-contract=off"] + NO_PACKED_FP32),
    ("cnorm.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _deps(src):
    return [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(PKG), "include", "mridc_amd.h")]


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, rebuilt = [], False
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIBDIR, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + COMMON + extra + os.environ.get("MRX_BUILD_DEFS", "").split() + ["-x", "hip", "-c", src, "-o", obj]
        stamp = obj + ".flags"      # the command the object was built with: a probe build (-DMRX_PROBE) never leaks into a product build
        same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_flags or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src)):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
