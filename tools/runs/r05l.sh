#!/bin/bash
# round 5, GPU call L: co-residency probe -- layer 1 on one stream, layer 2 on another: product kernels vs the small-footprint pair
O=gpurun_out/r05l; mkdir -p $O
for rep in 1 2; do
  timeout 300 python tools/probe/coresident.py 2>&1 | grep "layer 1 alone" | tee -a $O/cores.txt
  for v in cores cores2; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 python tools/probe/coresident.py 2>&1 | grep "layer 1 alone" | tee -a $O/cores.txt
  done
done
