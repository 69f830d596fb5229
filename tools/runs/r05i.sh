#!/bin/bash
# round 5, GPU call I: the folded statistics merge -- unit tests, model tests, E2EVN bench (fold on / off)
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py tests/test_gpu_backward.py tests/test_gpu_graph.py tests/test_gpu_concurrent_streams.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-300
for rep in 1 2; do
  for fold in 1 0; do
    MRX_UNET_FOLD=$fold timeout 300 python bench.py --model e2evn --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fold', $fold, 'e2evn', round(r['value'],1), 'slices/s', round(r['ms_per_step'],3), 'ms')" | tee -a $O/e2evn.txt
  done
done
