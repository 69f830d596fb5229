#!/bin/bash
# round 6, GPU call X: batch x streams of the secondary lines on lib 266 (no kernel change)
O=gpurun_out/r06x; mkdir -p $O
: > $O/sweep.txt
for bs in "8 2" "8 3" "16 2" "8 2"; do
  set -- $bs
  python3 bench.py --model e2evn --precision 16 --batch $1 --streams $2 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('e2evn16 batch $1 x streams $2:', round(r['value'],1))" >> $O/sweep.txt
done
for p in 32 16; do for st in 4 6 8 4; do
  python3 bench.py --model qcirim --precision $p --streams $st --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('qcirim precision $p streams $st:', round(r['value'],1))" >> $O/sweep.txt
done; done
cat $O/sweep.txt
