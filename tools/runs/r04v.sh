#!/bin/bash
O=gpurun_out/r04v; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_n4_models.py -q -x -k "coil_operator or e2evn or unet or varnet or vn" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_graph.py -q -x > $O/graph_$i.txt 2>&1; tail -1 $O/graph_$i.txt; done
MRIDC_AMD_LLG372_NO_Y=0 timeout 300 python -m pytest tests/test_gpu_graph.py -q -x > $O/graph_y.txt 2>&1; tail -1 $O/graph_y.txt
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 6 --warmup 2 > $O/bench_train_e2evn_$i.json 2> $O/bench_train_e2evn_$i.err; head -c 200 $O/bench_train_e2evn_$i.json; echo; done
