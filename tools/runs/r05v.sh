#!/bin/bash
# round 5, GPU call V: layer 2 with every barrier followed by MFMAs whose operands are in registers (PFETCH) -- bit-identity on the layer-2 tests, then A/B
O=gpurun_out/r05v; mkdir -p $O
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_pf/libmridc_amd.so timeout 900 python -m pytest tests/test_gpu_cb8.py tests/test_gpu_robust_f16.py tests/test_gpu_headline.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_pf.txt
for rep in 1 2 3; do
  for v in cur pf; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 python tools/probe/l2_time.py 2>&1 | tail -1 | tee -a $O/l2_time.txt
  done
done
