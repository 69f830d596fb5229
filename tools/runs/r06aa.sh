#!/bin/bash
# round 6, GPU call AA: bf16 training eager on two streams against one hipGraph replay per step (no kernel change)
O=gpurun_out/r06aa; mkdir -p $O
: > $O/ab.txt
for g in 0 1 0 1; do
  python3 bench.py --train --dtype bf16 --train-graph $g --no-cpu-baseline --no-other-configs --steps 8 --warmup 3 2>$O/err_$g.txt | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('train-graph $g:', round(r['value'],2), r['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/err_1.txt
