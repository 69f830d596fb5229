#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python tools/probe/lib_ab.py $PWD/mridc_amd/lib_pk_conv_bwd/libmridc_amd.so conv_bwd.hip > $O/lib_ab.txt 2>&1
cat $O/lib_ab.txt | tail -40
timeout 900 python -m pytest tests/test_gpu_train_bf16.py -q -s > $O/pytest_train_bf16.txt 2>&1
grep -E "passed|failed|^FAILED|Error|bf16-storage" $O/pytest_train_bf16.txt | head -30
