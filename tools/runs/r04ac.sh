#!/bin/bash
O=gpurun_out/r04ac; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_headline.py tests/test_gpu_backward.py -q -x -k "cell or tape or training or layer" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train_$i.json 2> $O/bench_train_$i.err; head -c 200 $O/bench_train_$i.json; echo; done
