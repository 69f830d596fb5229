#!/bin/bash
# one-stream kernel trace of the 2-D-mask line (4 slices per launch: every kernel alone on the chip)
O=gpurun_out/r04bg; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --mask 2d --streams 1 --steps 6 --warmup 2 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/mask2d_one_stream_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -12 $R/$O/mask2d_one_stream_kernel_stats.md | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof1 -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --mask 2d --streams 1 --batch 1 --steps 6 --warmup 2 > $R/$O/prof1.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof1/t_results.db > $R/$O/mask2d_b1_one_stream_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof1
head -9 $R/$O/mask2d_b1_one_stream_kernel_stats.md | cut -c1-160
