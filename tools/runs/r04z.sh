#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_cb8.py tests/test_gpu_graph.py tests/test_gpu_backward.py -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
for g in 1 0; do MRIDC_AMD_LLG372_GATHER=$g timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_g$g.json 2> $O/bench_g$g.err; python - $g <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04z/bench_g%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("gather", sys.argv[1], "headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "fft", d["roofline_fft"]["avg_ms"], d["roofline_fft"]["frac"], d["parity_vs_oracle"]["rel_l2"] if "parity_vs_oracle" in d else None)
print(d.get("breakdown_ms"))
PY
done
