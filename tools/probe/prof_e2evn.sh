#!/bin/bash
# rocprofv3 kernel trace of the E2EVN forward (eager, one stream): $1 = extra bench flags, $2 = output tag
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_e2evn_$2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o e -- python3 bench.py --model e2evn $1 --steps 4 --warmup 1 --graph 0 --streams 1 --no-cpu-baseline > $O/log 2>&1
python3 tools/rocpd_summary.py $O/prof/*results.db > $O/kernel_stats.md
rm -rf $O/prof
head -30 $O/kernel_stats.md | cut -c1-150; tail -2 $O/kernel_stats.md
