"""CPU oracle for the mridc reconstruction hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's unrolled-reconstruction hot
path.  It exists so the HIP kernels can be checked against
something that was itself pinned to the reference:

* pinned by `tests/golden/*.npz` -- vectors produced by importing the real reference leaf modules
  in the build container (`tests/golden/generate_golden.py`, `tests/golden/_refshim.py`), and by a
  restatement of the reference's own value-pinned tests (`tests/collections/reconstruction/test_fft.py`
  -> `tests/test_oracle_fft.py`, against numpy fp64).
* only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
  The product package `mridc_amd` never imports `oracle` and has no CPU fallback.

The third-party arithmetic the reference delegates to (torch.fft / torch.nn.functional.conv2d,
torch==1.12.0 pinned in the reference's requirements) is delegated to the same torch CPU kernels
here, so that the CPU baseline timed by bench.py is the reference's own arithmetic; an independent
numpy float64 DFT path (`oracle.fft.fft2_np64`) cross-checks it.

Every function cites the reference file:line it follows (paths relative to the reference root).
"""
from . import fft, utils, rim, unet, varnet, models, metrics, qrim, transforms, cascadenet, vsnet, dc_layers, rvn, dunet, amp  # noqa: F401
