#!/bin/bash
O=gpurun_out/r04bd; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04bd/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"))
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("avg_ms"), (v.get("roofline") or {}).get("frac"))
PY
