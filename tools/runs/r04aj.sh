#!/bin/bash
O=gpurun_out/r04aj; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04aj/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "traffic", d["roofline"]["traffic"], d["roofline_fft"]["traffic"], d["roofline_fft"]["gather_form"]["traffic"], "mfma", d["roofline"].get("mfma_util_pmc"))
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("traffic"))
PY
