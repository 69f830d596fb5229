#!/bin/bash
O=gpurun_out/r04w; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_n4_models.py -q -x -k "coil_operator or e2evn or unet or varnet or vn" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_graph.py -q -x > $O/graph_$i.txt 2>&1; tail -1 $O/graph_$i.txt; done
