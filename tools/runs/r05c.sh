#!/bin/bash
# round 5, GPU call C: layer-2 variants A/B (alternating, one process each)
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2 3; do
  for v in ${L2_VARIANTS:-base diet}; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 python tools/probe/l2_time.py 2>&1 | tail -1 | tee -a $O/l2_time.txt
  done
done
