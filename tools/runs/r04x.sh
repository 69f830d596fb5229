#!/bin/bash
O=gpurun_out/r04x; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -k "graphed or e2evn" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 6 --warmup 2 > $O/bench_train_e2evn_$i.json 2> $O/bench_train_e2evn_$i.err; head -c 200 $O/bench_train_e2evn_$i.json; echo; tail -2 $O/bench_train_e2evn_$i.err | cut -c1-300; done
