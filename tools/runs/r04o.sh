#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_models.py -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 4 --warmup 1 > $O/bench_train_e2evn.json 2> $O/bench_train_e2evn.err; head -c 200 $O/bench_train_e2evn.json; echo
timeout 600 python bench.py --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train_checks.json 2> $O/bench_train_checks.err; python -c "
import json
d=json.loads(open('$O/bench_train_checks.json').read().strip().splitlines()[-1]); print(d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('parity_vs_oracle',{}).get('within_tolerance'))"
