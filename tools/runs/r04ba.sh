#!/bin/bash
O=gpurun_out/r04ba; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_cb8.py tests/test_gpu_graph.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ba/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["breakdown_ms"]["llg"], d["roofline_fft"]["avg_ms"], d["roofline_fft"]["gather_form"]["avg_ms"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 10 --warmup 2 --streams 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/headline_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -9 $R/$O/headline_kernel_stats.md | cut -c1-150
