#!/bin/bash
# round 4, GPU call D: bisect on the FAILING weights (seed 5, boosted), unit tests of the bf16-storage kernels
O=gpurun_out/r04d; mkdir -p $O
for v in "" conv_bwd rim_layer rim_layer_wino conv conv_bf16; do
  if [ -n "$v" ]; then export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_pk_$v/libmridc_amd.so; else unset MRIDC_AMD_LIB; fi
  echo "=== packed-fp32 allowed in: ${v:-nothing}" >> $O/bisect.txt
  timeout 300 python tools/probe/train_parity.py 4 48 40 f32 >> $O/bisect.txt 2>&1
done
unset MRIDC_AMD_LIB
grep -E "===|whole|seed" $O/bisect.txt
timeout 900 python -m pytest tests/test_gpu_train_bf16.py -q -s -x > $O/pytest_train_bf16.txt 2>&1
tail -60 $O/pytest_train_bf16.txt
