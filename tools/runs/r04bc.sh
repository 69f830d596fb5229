#!/bin/bash
O=gpurun_out/r04bc; mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_n4_models.py -m gpu -q -k "dunet" > $O/dunet$i.txt 2>&1; tail -1 $O/dunet$i.txt; done
timeout 900 python -m pytest tests/test_n4_models.py tests/test_preprocessing.py tests/test_gpu_unet_fused.py -m gpu -q > $O/rest.txt 2>&1; tail -3 $O/rest.txt | cut -c1-200
