#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_cb8.py -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04t/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "parity", d["parity_vs_oracle"], "fft", d["roofline_fft"]["avg_ms"], d["roofline_fft"]["frac"])
print(d.get("breakdown_ms"))
PY
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train.json 2> $O/bench_train.err; head -c 200 $O/bench_train.json; echo
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype f32 --steps 2 --warmup 1 > $O/bench_train_f32.json 2> $O/bench_train_f32.err; head -c 200 $O/bench_train_f32.json; echo
