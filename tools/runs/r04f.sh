#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
timeout 300 python tools/probe/tl_determinism.py > $O/determinism.txt 2>&1; cat $O/determinism.txt | tail -12
timeout 300 python tools/probe/train_parity.py 4 48 40 bf16 > $O/parity_small.txt 2>&1; cat $O/parity_small.txt | tail -30
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16.py tests/test_gpu_train_bf16.py -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
