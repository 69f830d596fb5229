"""mridc_amd -- MI355X (gfx950) implementation of the mridc unrolled-reconstruction hot path.

Same module paths / names / signatures as `mridc.collections.{common.parts, reconstruction.models}` for the
path in SURVEY.md section 8; every operator dispatches to hand-written HIP kernels in
`mridc_amd/lib/libmridc_amd.so` (C ABI: include/mridc_amd.h).  No CPU fallback.
"""
__version__ = "0.1.0"
