#!/bin/bash
# round 5, GPU call N: pooled zero scalars -- GPU suite, qCIRIM / RIM-GRU lines
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt | cut -c1-200
for rep in 1 2; do
  timeout 300 python bench.py --model qcirim --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qcirim', round(r['value'],1), 'slices/s', round(r['ms_per_step'],3), 'ms')" | tee -a $O/q.txt
done
timeout 300 python bench.py --rnn GRU --cascades 1 --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rim-gru', round(r['value'],1), 'slices/s')" | tee -a $O/q.txt
