#!/bin/bash
O=gpurun_out/r04at; mkdir -p $O
for m in "2d 1" "2d 4" "2d 8"; do
  set -- $m
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --mask $1 --batch $2 > $O/m$1b$2.json 2> $O/m$1b$2.err
  python - $O/m$1b$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("mask", sys.argv[2], "batch", sys.argv[3], "->", round(d["value"], 2), "slices/s")
except Exception as e:
    print("mask", sys.argv[2], "batch", sys.argv[3], "failed", e)
PY
done
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04at/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["config"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "parity", d["parity_vs_oracle"]["rel_l2"])
print(d["breakdown_ms"]); print(d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline"]["traffic"], d["roofline_fft"]["frac"], d["roofline_fft"]["avg_ms"])
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), v.get("error"))
PY
