#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_bf16.py -q -k "cell_backward" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train.json 2> $O/bench_train.err; head -c 250 $O/bench_train.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $O/prof_train/t_results.db | head -22 | cut -c1-150
