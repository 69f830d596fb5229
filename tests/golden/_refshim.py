"""Import shim for the *reference* leaf modules (only usable where /root/reference exists).

Used ONLY by tests/golden/generate_golden.py in the build container to produce the committed
golden vectors.  Nothing here is imported by the product, by the GPU tests or by bench.py, and
no reference source is copied: the shim registers empty parent packages so the reference's
package __init__ files (which pull pytorch_lightning / omegaconf / h5py) never execute, then
imports the leaf modules from where they lie.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("MRIDC_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def install():
    if not os.path.isdir(os.path.join(REF, "mridc")):
        raise RuntimeError(f"reference tree not found at {REF}")
    _stub("omegaconf", ListConfig=type("ListConfig", (list,), {}), DictConfig=type("DictConfig", (dict,), {}),
          OmegaConf=type("OmegaConf", (), {}))
    _stub("h5py")
    _stub("numba", jit=lambda *a, **k: (lambda f: f))
    pkgs = [
        "mridc", "mridc.collections", "mridc.collections.common", "mridc.collections.common.parts",
        "mridc.collections.common.losses",
        "mridc.collections.reconstruction", "mridc.collections.reconstruction.data", "mridc.collections.reconstruction.parts",
        "mridc.collections.reconstruction.models", "mridc.collections.reconstruction.models.rim",
        "mridc.collections.reconstruction.models.varnet", "mridc.collections.reconstruction.models.unet_base",
        "mridc.collections.reconstruction.models.conv", "mridc.collections.reconstruction.models.cascadenet",
        "mridc.collections.reconstruction.models.variablesplittingnet", "mridc.collections.reconstruction.models.sigmanet",
        "mridc.collections.reconstruction.models.recurrentvarnet", "mridc.collections.reconstruction.models.didn",
        "mridc.collections.quantitative", "mridc.collections.quantitative.models",
        "mridc.collections.quantitative.models.qrim",
    ]
    for p in pkgs:
        if p not in sys.modules:
            m = types.ModuleType(p)
            m.__path__ = [os.path.join(REF, *p.split("."))]
            sys.modules[p] = m


def load(name):
    install()
    return importlib.import_module(name)
