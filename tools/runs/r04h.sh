#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
timeout 300 python tools/probe/train_parity.py 4 48 40 bf16 --cascades 2 --slice 2 > $O/parity_2c.txt 2>&1; grep -E "seed|whole|cirim.1" $O/parity_2c.txt | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_backward.py tests/test_gpu_bf16.py -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k training > $O/pytest_headline_train.txt 2>&1; grep -E "training parity|passed|failed|Error" $O/pytest_headline_train.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench_train_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_train_prof.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $O/prof_train/*/*.db 2>/dev/null | head -34 | cut -c1-160 || python tools/rocpd_summary.py $O/prof_train/t_results.db | head -34 | cut -c1-160
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train.json 2> $O/bench_train.err; head -c 300 $O/bench_train.json
