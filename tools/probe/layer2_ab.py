"""A/B in one process: second RIM layer (two-term fp16 route) with MRX_L2_ABL variants interleaved (library built with -DMRX_PROBE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mridc_amd import ops
from tools.probe.layer2_ablate2 import setup, timed  # noqa: E402
variants = [int(v) for v in os.environ.get("AB_VARIANTS", "0,1024").split(",")]
fn = setup(640, 372)
for rep in range(4):
    for abl in variants:
        if abl:
            os.environ["MRX_L2_ABL"] = str(abl)
        else:
            os.environ.pop("MRX_L2_ABL", None)
        print("rep %d ABL %4d %.2f us" % (rep, abl, timed(fn, 200)), flush=True)
