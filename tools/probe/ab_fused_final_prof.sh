#!/bin/bash
# kernel durations of the RIM step with and without the final convolution fused into layer 2 (rocprofv3 kernel trace, eager, one stream)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for ff in 0 1; do
  O=gpurun_out/ab_ff$ff
  rm -rf $O; mkdir -p $O
  export MRIDC_AMD_FUSED_FINAL=$ff
  rocprofv3 --kernel-trace --stats -d $O/prof -o h -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --graph 0 --streams 1 > $O/log 2>&1
  python3 tools/rocpd_summary.py $O/prof/*results.db > $O/kernel_stats.md
  rm -rf $O/prof
  echo "== fused $ff"; head -14 $O/kernel_stats.md
done
