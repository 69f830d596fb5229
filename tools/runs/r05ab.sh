#!/bin/bash
# round 5, GPU call AB: HIP backward of the gates, group norm / un-norm and zero padding -- tests, then the E2EVN training line before / after is in bench (train_e2evn)
O=gpurun_out/r05ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 6 --warmup 2 2> $O/err.txt | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('e2evn train', r['value'], r['ms_per_step'])" | tee -a $O/train_e2evn.txt; done
