#!/bin/bash
# batch x streams scan of the headline on the round-4 kernels (480 tiles per slice on 256 CUs: 8 slices per launch = exactly 15 rounds)
O=gpurun_out/r04as; mkdir -p $O
for bs in "1 2" "8 1" "4 2" "8 2" "2 2" "16 1"; do
  set -- $bs
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 > $O/b$1s$2.json 2> $O/b$1s$2.err
  python - $O/b$1s$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("batch", sys.argv[2], "streams", sys.argv[3], "->", round(d["value"], 2), "slices/s", d["breakdown_ms"]["llg"], d["breakdown_ms"]["conv_layer1"], d["breakdown_ms"]["conv_layer2"])
except Exception as e:
    print("batch", sys.argv[2], "streams", sys.argv[3], "failed", e)
PY
done
