#!/bin/bash
O=gpurun_out/r04ae; mkdir -p $O
timeout 1500 python tools/probe/train_parity.py > $O/train_parity_lib244.txt 2>&1; cat $O/train_parity_lib244.txt | grep -v amdgpu.ids
