#!/bin/bash
# round 5, GPU call Y: kernel trace of the E2EVN training step (where do its ~1500 torch launches go)
O=gpurun_out/r05y; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --model e2evn --steps 3 --warmup 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/train_e2evn_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -70 $R/$O/train_e2evn_kernel_stats.md | cut -c1-170
tail -3 $R/$O/prof.log | cut -c1-300
