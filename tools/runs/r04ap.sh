#!/bin/bash
O=gpurun_out/r04ap; mkdir -p $O
for i in 1 2; do timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_$i.txt 2>&1; tail -2 $O/pytest_$i.txt | cut -c1-200; done
