#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
timeout 300 python tools/probe/train_parity.py 4 48 40 bf16 --cascades 2 --slice 2 > $O/parity_2c.txt 2>&1; cat $O/parity_2c.txt | tail -56
timeout 300 python tools/probe/train_parity.py 4 48 40 bf16 --cascades 1 --slice 2 > $O/parity_1c.txt 2>&1; grep -E "seed|whole" $O/parity_1c.txt
