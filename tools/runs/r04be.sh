#!/bin/bash
# one-stream kernel trace of the bf16 training step (every kernel alone on the chip) + the step time of two plain runs
O=gpurun_out/r04be; mkdir -p $O; R=$GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train_$i.json 2> $O/bench_train_$i.err; head -c 200 $O/bench_train_$i.json; echo; done
cd /tmp && export TMPDIR=/tmp
export MRIDC_AMD_TL_SIDE_STREAM=0
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/train_serial_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -16 $R/$O/train_serial_kernel_stats.md | cut -c1-150
