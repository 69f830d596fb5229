#!/bin/bash
# E2EVN batch x streams scan on the round-4 kernels
O=gpurun_out/r04av; mkdir -p $O
for bs in "4 2" "8 2" "8 1" "16 1" "6 2" "12 1" "4 3"; do
  set -- $bs
  timeout 300 python bench.py --model e2evn --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 > $O/b$1s$2.json 2> $O/b$1s$2.err
  python - $O/b$1s$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("e2evn batch", sys.argv[2], "streams", sys.argv[3], "->", round(d["value"], 1), "slices/s")
except Exception as e:
    print("e2evn batch", sys.argv[2], "streams", sys.argv[3], "failed", e)
PY
done
for bs in "1 2" "2 2" "4 2" "2 1" "4 1"; do
  set -- $bs
  timeout 300 python bench.py --model qcirim --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 > $O/q$1s$2.json 2> $O/q$1s$2.err
  python - $O/q$1s$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("qcirim batch", sys.argv[2], "streams", sys.argv[3], "->", round(d["value"], 1), "slices/s")
except Exception as e:
    print("qcirim batch", sys.argv[2], "streams", sys.argv[3], "failed", e)
PY
done
