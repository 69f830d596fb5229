#!/bin/bash
# round 5, GPU call X: layer 2 (FAST) with its memory streams made cache-resident / dropped, one at a time (timing variants, garbage results): which of them the kernel waits for
O=gpurun_out/r05x; mkdir -p $O
for rep in 1 2; do
  for v in ${L2_VARIANTS:-cur abl_XHOT abl_NOST abl_HPHOT abl_ALL}; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 python tools/probe/l2_time.py 2>&1 | tail -1 | tee -a $O/l2_time.txt
  done
done
