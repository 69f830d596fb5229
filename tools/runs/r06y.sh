#!/bin/bash
# round 6, GPU call Y: batch x streams of the 2-D-mask and precision-16 CIRIM lines on lib 266 (no kernel change)
O=gpurun_out/r06y; mkdir -p $O
: > $O/sweep.txt
for bs in "4 2" "4 3" "2 4" "6 2" "4 2"; do
  set -- $bs
  python3 bench.py --mask 2d --batch $1 --streams $2 --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('mask2d batch $1 x streams $2:', round(r['value'],1))" >> $O/sweep.txt
done
for bs in "8 2" "8 3" "16 2" "8 2"; do
  set -- $bs
  python3 bench.py --precision 16 --batch $1 --streams $2 --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('precision16 batch $1 x streams $2:', round(r['value'],1))" >> $O/sweep.txt
done
cat $O/sweep.txt
