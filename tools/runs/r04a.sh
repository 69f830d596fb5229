#!/bin/bash
# round 4, GPU call A: stand-alone packed-fp32 / MFMA reproducer, bf16 training tape vs the three oracle arithmetics, A/B of the library built
# without packed-fp32 instructions anywhere
O=gpurun_out/r04a; mkdir -p $O
timeout 600 ./tools/probe/pk_mfma_repro 6 > $O/pk_repro.txt 2>&1
timeout 900 python tools/probe/train_parity.py > $O/train_parity.txt 2>&1
NOPK=$PWD/mridc_amd/lib_nopk/libmridc_amd.so
for v in base nopk; do
  if [ $v = nopk ]; then export MRIDC_AMD_LIB=$NOPK; else unset MRIDC_AMD_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 20 --warmup 3 > $O/bench_cirim_$v.json 2> $O/bench_cirim_$v.err
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 20 --warmup 3 > $O/bench_e2evn_$v.json 2> $O/bench_e2evn_$v.err
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model qcirim --steps 20 --warmup 3 > $O/bench_qcirim_$v.json 2> $O/bench_qcirim_$v.err
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > $O/bench_train_$v.json 2> $O/bench_train_$v.err
done
export MRIDC_AMD_LIB=$NOPK
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_host_logic.py > $O/pytest_nopk.txt 2>&1
tail -5 $O/pytest_nopk.txt
grep -h '"value"' $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['metric'][:60], round(d['value'], 2))
    except Exception as e: print('?', l[:100])
"
tail -30 $O/pk_repro.txt
cat $O/train_parity.txt | tail -40
