#!/bin/bash
O=gpurun_out/r04ag; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 4 --warmup 1 --graph 0 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/train_e2evn_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -45 $R/$O/train_e2evn_kernel_stats.md | cut -c1-160
