"""Build libmridc_amd.so (HIP, gfx950) in-tree with hipcc.  Called by __graft_entry__.build() and `python -m mridc_amd._build`."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmridc_amd.so")
CHECK_LIBDIR = os.path.join(PKG, "lib_chk")     # the CHECK build (-DMRX_CHECK_BOUNDS, include/mridc_amd.h: mrx_checks_enabled): every operand bound verified

# NO packed-fp32 vector instructions (v_pk_add / mul / fma_f32) anywhere in the library.  On MI355X a wave executing v_pk_*_f32 with op_sel / neg
# operand modifiers (the forms complex arithmetic needs, and the ones the compiler forms for it) returns WRONG results while a wave of another kernel
# on the same SIMD issues v_mfma_f32_16x16x32_f16 -- reproduced stand-alone, without this library, by tools/probe/pk_mfma_repro.hip (round 4:
# 442 427 wrong values in 4 875 of 12 288 workgroup-runs for exactly that pair, bit-exact in the 100 other victim / aggressor cells).
# Found in round 3 on two streams (the transforms of one slice next to the U-Net convolutions of another).  The target feature stops the compiler
# from forming packed-fp32 instructions, MRX_NO_PACKED_FP32 selects the scalar complex layer of pfa372.h instead of its inline assembly;
# tests/test_host_logic.py disassembles every object of the library and counts them.  It is also the faster build (packed fp32 is two passes on
# this chip and an anti-lever beside MFMAs): headline 121.4 -> 128.6 slices/s on the same box (gpurun_out/r04a).
NO_PACKED_FP32 = ["-DMRX_NO_PACKED_FP32", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# (source, extra flags).  elementwise.hip / qmri.hip are built without fp contraction so that the pointwise complex
# operators round exactly like the reference's separate torch ops (mul, mul, sub).
SOURCES = [
    ("api.cpp", []),
    ("host_masks.cpp", []),
    ("fft.hip", []),
    ("llg372.hip", ["-fno-slp-vectorize"]),
    ("elementwise.hip", ["-ffp-contract=off"]),
    ("conv.hip", []),
    ("rim_layer.hip", []),
    ("rim_layer_wino.hip", []),
    ("rim_layer1_sb.hip", []),
    ("rim_layer2_sb.hip", []),
    ("rim_amp16.hip", []),
    ("gated_cell.hip", []),
    ("gated_cell_sb.hip", []),
    ("conv_bwd.hip", []),
    ("conv_bf16.hip", []),
    ("conv_sbs.hip", []),
    ("unet.hip", []),
    ("unet_fused.hip", []),
    ("unet_f16.hip", []),
    ("qmri.hip", ["-ffp-contract=off"]),
    ("cnorm.hip", []),
    ("train_bf16.hip", []),
    ("diff_bwd.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + NO_PACKED_FP32


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _deps(src):
    return [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(PKG), "include", "mridc_amd.h")]


def build(force=False, verbose=True, libdir=None, extra_all=(), packed_ok=()):
    """`libdir` / `extra_all` / `packed_ok`: a second build of the library in its own directory (A/B runs through MRIDC_AMD_LIB) with extra flags
    for every source, or with the sources named in `packed_ok` compiled WITHOUT the no-packed-fp32 flags (bisecting what that switch changes)."""
    libdir = LIBDIR if libdir is None else libdir
    lib = os.path.join(libdir, "libmridc_amd.so")
    os.makedirs(libdir, exist_ok=True)
    hipcc = _hipcc()
    objs, rebuilt = [], False
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(libdir, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        common = [f for f in COMMON if f not in NO_PACKED_FP32] if name in packed_ok else COMMON
        cmd = [hipcc] + common + extra + [f for f in extra_all if f not in extra] + os.environ.get("MRX_BUILD_DEFS", "").split() + ["-x", "hip", "-c", src, "-o", obj]
        stamp = obj + ".flags"      # the command the object was built with: a probe build (-DMRX_PROBE) never leaks into a product build
        same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_flags or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src)):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
            rebuilt = True
    if rebuilt or not os.path.exists(lib):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--packed-ok" in sys.argv:           # python -m mridc_amd._build --packed-ok conv_bwd.hip,rim_layer.hip  -> mridc_amd/lib_pk_conv_bwd+rim_layer/
        names = tuple(sys.argv[sys.argv.index("--packed-ok") + 1].split(","))
        print(build(libdir=os.path.join(PKG, "lib_pk_" + "+".join(os.path.splitext(n)[0] for n in names)), packed_ok=names))
    elif "--check-bounds" in sys.argv:      # python -m mridc_amd._build --check-bounds -> mridc_amd/lib_chk/ (MRIDC_AMD_LIB=.../lib_chk/libmridc_amd.so)
        print(build(libdir=CHECK_LIBDIR, extra_all=["-DMRX_CHECK_BOUNDS"]))
    else:
        print(build(force="--force" in sys.argv))
