#!/bin/bash
# the tap gather folded into the next gradient launch (MRIDC_AMD_LLG372_GATHER=1) against its own launch (=0), by batch size
O=gpurun_out/r04aw; mkdir -p $O
for bs in "8 1" "8 0" "4 1" "4 0" "2 1" "2 0" "8 1" "8 0"; do
  set -- $bs
  MRIDC_AMD_LLG372_GATHER=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams 2 > $O/b$1g$2.json 2> $O/b$1g$2.err
  python - $O/b$1g$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("batch", sys.argv[2], "fold", sys.argv[3], "->", round(d["value"], 2), "slices/s", d["breakdown_ms"]["llg"], d["breakdown_ms"]["final"])
except Exception as e:
    print("batch", sys.argv[2], "fold", sys.argv[3], "failed", e)
PY
done
for bs in "8 2" "12 2" "16 2" "8 3" "10 2"; do
  set -- $bs
  timeout 300 python bench.py --model e2evn --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 > $O/e$1s$2.json 2> $O/e$1s$2.err
  python - $O/e$1s$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("e2evn batch", sys.argv[2], "streams", sys.argv[3], "->", round(d["value"], 1), "slices/s")
except Exception as e:
    print("e2evn batch", sys.argv[2], "streams", sys.argv[3], "failed", e)
PY
done
