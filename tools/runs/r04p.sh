#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_headline.py -q -k "cell or tape or training" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt | cut -c1-250
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train_$i.json 2> $O/bench_train_$i.err; head -c 200 $O/bench_train_$i.json; echo; done
