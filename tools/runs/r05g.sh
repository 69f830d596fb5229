#!/bin/bash
# round 5, GPU call G: GPU suite + bench on the FAST layer-2 route (product build)
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-300
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r05g/bench.json') if l.startswith('{')][-1]); print(json.dumps(r['summary'])); print(r['breakdown_ms'])
PY
