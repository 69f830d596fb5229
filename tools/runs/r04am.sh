#!/bin/bash
python tools/probe/wgrad_time.py 2>&1 | tail -1
for v in 1 2 3; do MRIDC_AMD_LIB=$GRAFT_REPO_ROOT/mridc_amd/lib_abl$v/libmridc_amd.so python tools/probe/wgrad_time.py 2>&1 | tail -1; done
