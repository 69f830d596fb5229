#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_bf16.py -q -k "cell_backward or weight_gradient" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt | cut -c1-250
timeout 900 python tools/probe/train_parity.py > $O/parity_headline.txt 2>&1; cat $O/parity_headline.txt | cut -c1-200
