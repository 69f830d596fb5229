#!/bin/bash
# round 6, GPU call J: E2EVN-6 and qCIRIM against (slices per launch, streams): do the U-Net's activations (109 MB per tensor at 8 slices) pass from producer to consumer
# through the 256 MB memory-side cache at smaller batches?
O=gpurun_out/r06j; mkdir -p $O
: > $O/sweep.txt
for cfg in "--batch 8 --streams 2" "--batch 4 --streams 2" "--batch 4 --streams 4" "--batch 2 --streams 4" "--batch 8 --streams 3" "--batch 16 --streams 1" "--batch 4 --streams 3" "--batch 8 --streams 2"; do
  python3 bench.py --model e2evn --no-cpu-baseline --steps 8 --warmup 2 $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('e2evn $cfg', round(r['value'],1), round(r['ms_per_step'],3))" >> $O/sweep.txt
done
for cfg in "--batch 1 --streams 2" "--batch 2 --streams 2" "--batch 1 --streams 4" "--batch 4 --streams 2" "--batch 2 --streams 4"; do
  python3 bench.py --model qcirim --no-cpu-baseline --steps 10 --warmup 2 $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('qcirim $cfg', round(r['value'],1), round(r['ms_per_step'],3))" >> $O/sweep.txt
done
cat $O/sweep.txt
