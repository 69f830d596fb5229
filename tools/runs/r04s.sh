#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O; R=$GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 6 --warmup 2 > $O/bench_train_e2evn_$i.json 2> $O/bench_train_e2evn_$i.err; head -c 200 $O/bench_train_e2evn_$i.json; echo; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 3 --warmup 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/train_e2evn_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -40 $R/$O/train_e2evn_kernel_stats.md | cut -c1-150
