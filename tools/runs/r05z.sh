#!/bin/bash
# round 5, GPU call Z: kernel trace of the bf16 CIRIM training step on ONE stream (each kernel's own duration: in the two-stream trace a side-stream launch's
# duration includes its wait for CUs)
O=gpurun_out/r05z; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MRIDC_AMD_TL_SIDE_STREAM=0
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/train_bf16_one_stream_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -34 $R/$O/train_bf16_one_stream_kernel_stats.md | cut -c1-150
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $R/$O/prof.log | head -3
