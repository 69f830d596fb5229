#!/bin/bash
# round 5, GPU call Q: CIRIM bf16 training step eager (two streams) vs one hipGraph replay, today's library
O=gpurun_out/r05q; mkdir -p $O
for rep in 1 2; do
  for g in 0 1; do
    timeout 400 python bench.py --train --dtype bf16 --steps 4 --warmup 2 --no-other-configs --no-stream-inputs --no-cpu-baseline --train-graph $g 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train-graph', $g, round(r['value'],2), 'slices/s', round(r['ms_per_step'],2), 'ms')" | tee -a $O/train.txt
  done
done
