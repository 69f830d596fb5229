#!/bin/bash
export MRIDC_AMD_LIB=$GRAFT_REPO_ROOT/mridc_amd/lib_probe/libmridc_amd.so
for v in 0 1 2 4 8 16 3 7 24 31; do MRX_UCONVH_ABLATE=$v python tools/probe/uconv_h_ablate.py 2>&1 | tail -1; done
