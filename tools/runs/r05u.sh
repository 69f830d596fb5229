#!/bin/bash
# round 5, GPU call U: weight gradients summed over a cascade's time-steps in the workgroups' slots (training.TL_WGRAD_SERIES) -- tests, then the bf16 training line A/B
O=gpurun_out/r05u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  for v in 0 1; do
    MRIDC_AMD_TL_WGRAD_SERIES=$v timeout 600 python bench.py --train --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2> $O/err_$v.txt | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('series=$v', r['value'], r['unit'], r['ms_per_step'], r.get('parity', {}).get('within_tolerance') if isinstance(r.get('parity'), dict) else None)" | tee -a $O/train.txt
  done
done
