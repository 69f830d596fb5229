"""A variant of libmridc_amd.so for A/B runs: ONE source recompiled (another file, or the tree's file with extra -D flags), linked against the other
objects of the product build in mridc_amd/lib/.

  python tools/probe/build_variant.py NAME rim_layer2_sb.hip [--src /path/to/other_version.hip] [-DMRX_L2_XYZ ...]   ->  mridc_amd/lib_v_NAME/libmridc_amd.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mridc_amd import _build  # noqa: E402


def main():
    name, target = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    src = os.path.join(_build.CSRC, target)
    if "--src" in rest:
        i = rest.index("--src")
        src = rest[i + 1]
        rest = rest[:i] + rest[i + 2:]
    _build.build(verbose=False)                                     # the product objects are current
    out = os.path.join(_build.PKG, "lib_v_" + name)
    os.makedirs(out, exist_ok=True)
    extra = dict(_build.SOURCES)[target]
    obj = os.path.join(out, os.path.splitext(target)[0] + ".o")
    cmd = [_build._hipcc()] + _build.COMMON + extra + rest + ["-I", _build.CSRC, "-x", "hip", "-c", src, "-o", obj]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    objs = [obj if n == target else os.path.join(_build.LIBDIR, os.path.splitext(n)[0] + ".o") for n, _ in _build.SOURCES]
    lib = os.path.join(out, "libmridc_amd.so")
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
