#!/bin/bash
# the model-zoo lines of README (RIM-GRU / MGU, RecurrentVarNet, CascadeNet, VSNet) on the final library
O=gpurun_out/r04bf; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs $2 > $O/$1.json 2> $O/$1.err; python - $O/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "->", round(d["value"], 1), "slices/s")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run gru "--rnn GRU --cascades 1"
run mgu "--rnn MGU --cascades 1"
run rvn "--model rvn"
run ccnn "--model ccnn"
run vsnet "--model vsnet"
run e2evn18 "--model e2evn --unet 18x4"
