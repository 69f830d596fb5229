#!/bin/bash
# round 5, GPU call T: layer-2 software-pipelined 1x1 stage A/B + bit-identity of the variant on the layer-2 tests
O=gpurun_out/r05t; mkdir -p $O
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_swp/libmridc_amd.so timeout 600 python -m pytest tests/test_gpu_cb8.py tests/test_gpu_robust_f16.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_swp.txt
for rep in 1 2 3; do
  for v in cur swp; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 python tools/probe/l2_time.py 2>&1 | tail -1 | tee -a $O/l2_time.txt
  done
done
