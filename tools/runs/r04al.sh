#!/bin/bash
O=gpurun_out/r04al; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_headline.py tests/test_gpu_ops.py -q -x -k "varnet or e2evn or vn or expand or reduce or sens or llg" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-250
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 6 --warmup 2 > $O/bench_e2evn_$i.json 2> $O/bench_e2evn_$i.err; head -c 200 $O/bench_e2evn_$i.json; echo; done
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --mask 2d --steps 8 --warmup 2 > $O/bench_2d.json 2> $O/bench_2d.err; head -c 200 $O/bench_2d.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/e2evn_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -10 $R/$O/e2evn_kernel_stats.md | cut -c1-150
