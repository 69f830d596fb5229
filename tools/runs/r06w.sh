#!/bin/bash
# round 6, GPU call W: the headline's batch x streams on lib 266 (no kernel change): is 8 x 2 still the optimum?
O=gpurun_out/r06w; mkdir -p $O
: > $O/sweep.txt
for bs in "8 2" "8 3" "8 4" "8 2" "16 1" "16 2" "4 4" "8 3"; do
  set -- $bs
  python3 bench.py --batch $1 --streams $2 --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('batch $1 x streams $2:', round(r['value'],2), 'slices/s', round(r['ms_per_step'],2), 'ms/step')" >> $O/sweep.txt
done
cat $O/sweep.txt
