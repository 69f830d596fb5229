#!/bin/bash
# round 4, GPU call B: reproducer v2 (which operand modifiers / which MFMA shapes), the whole GPU suite on the library built without packed-fp32
# instructions anywhere (lib 237), headline kernel trace
O=gpurun_out/r04b; mkdir -p $O
timeout 900 ./tools/probe/pk_mfma_repro 4 > $O/pk_repro_v2.txt 2>&1
grep -n "MISMATCH\|cells with" $O/pk_repro_v2.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
ls -R $O/prof | head; python tools/rocprof_summary.py $O/prof 2>/dev/null | head -20
