#!/bin/bash
# round 5, GPU call H: cycle stamps of the FAST layer-2 route (probe build)
O=gpurun_out/r05h; mkdir -p $O
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_probe/libmridc_amd.so timeout 300 python tools/probe/l2_trace.py > $O/trace.txt 2>&1; grep "l2sb-trace" $O/trace.txt | tail -3
