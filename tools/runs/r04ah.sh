#!/bin/bash
O=gpurun_out/r04ah; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_unet_fused.py tests/test_gpu_models.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-250
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 6 --warmup 2 > $O/bench_train_e2evn_$i.json 2> $O/bench_train_e2evn_$i.err; head -c 200 $O/bench_train_e2evn_$i.json; echo; done
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 6 --warmup 2 > $O/bench_e2evn.json 2> $O/bench_e2evn.err; head -c 200 $O/bench_e2evn.json; echo
