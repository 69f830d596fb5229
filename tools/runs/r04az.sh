#!/bin/bash
O=gpurun_out/r04az; mkdir -p $O
for bs in "8 2 1d" "8 3 1d" "4 3 1d" "4 4 1d" "8 2 1d" "4 2 2d" "4 3 2d" "2 4 2d"; do
  set -- $bs
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 --mask $3 > $O/b$1s$2$3.json 2> $O/b$1s$2$3.err
  python - $O/b$1s$2$3.json $1 $2 $3 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("mask", sys.argv[4], "batch", sys.argv[2], "streams", sys.argv[3], "->", round(d["value"], 2), "slices/s")
except Exception as e:
    print("mask", sys.argv[4], "batch", sys.argv[2], "streams", sys.argv[3], "failed", e)
PY
done
