#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04u/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "fft", d["roofline_fft"]["avg_ms"], d["roofline_fft"]["frac"], d["roofline_fft"]["executed_frac"])
print(d.get("breakdown_ms")); print({k: v for k, v in d.items() if "parity" in k})
PY
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --mask 2d > $O/bench2d.json 2> $O/bench2d.err; head -c 150 $O/bench2d.json; echo
