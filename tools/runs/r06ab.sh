#!/bin/bash
# round 6, GPU call AB: the headline's two streams at equal priority against one high + one normal (no kernel change)
O=gpurun_out/r06ab; mkdir -p $O
: > $O/ab.txt
for p in eq hi-lo eq hi-lo; do
  MRX_BENCH_STREAM_PRIO=$p python3 bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('priorities $p:', round(r['value'],2))" >> $O/ab.txt
done
cat $O/ab.txt
