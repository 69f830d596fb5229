#!/bin/bash
O=gpurun_out/r04aa; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_cb8.py tests/test_gpu_graph.py tests/test_gpu_backward.py -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 10 --warmup 2 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/headline_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -12 $R/$O/headline_kernel_stats.md | cut -c1-150
