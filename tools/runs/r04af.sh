#!/bin/bash
O=gpurun_out/r04af; mkdir -p $O
for cfg in "1 2" "2 1" "2 2" "1 3" "4 1"; do set -- $cfg; timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --batch $1 --streams $2 --steps 10 --warmup 2 > $O/b$1_s$2.json 2> $O/b$1_s$2.err; python - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04af/b%s_s%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("batch", sys.argv[1], "streams", sys.argv[2], "->", round(d["value"], 2), "slices/s", d["breakdown_ms"]["conv_layer2"], d["breakdown_ms"]["conv_layer1"], d["breakdown_ms"]["llg"])
except Exception as ex:
    print("batch", sys.argv[1], "streams", sys.argv[2], "failed", ex)
PY
done
