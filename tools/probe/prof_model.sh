#!/bin/bash
# rocprofv3 kernel trace of one of bench.py's other models (eager, one stream): $1 = model flags, $2 = tag
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o e -- python3 bench.py $1 --steps 4 --warmup 1 --graph 0 --streams 1 --no-cpu-baseline > $O/log 2>&1
python3 tools/rocpd_summary.py $O/prof/*results.db > $O/kernel_stats.md
rm -rf $O/prof
head -16 $O/kernel_stats.md | cut -c1-150; tail -1 $O/kernel_stats.md
