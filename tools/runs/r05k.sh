#!/bin/bash
# round 5, GPU call K: the oracle's thread count in the bench's own context (16 vs 32 vs 8), the forced-dist single-rank path
O=gpurun_out/r05k; mkdir -p $O
for t in 16 32 8 16 32; do
  MRX_ORACLE_THREADS=$t timeout 400 python bench.py --steps 2 --warmup 1 --no-other-configs --no-stream-inputs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['cpu_baseline']; print('threads', c['cores'], 'cpu', round(c['value'],4), 'slices/s', [round(x,2) for x in c['sec_per_slice']], 'gpu', round(r['value'],1))" | tee -a $O/threads.txt
done
MRX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-other-configs --no-stream-inputs --no-cpu-baseline 2>$O/dist.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('forced dist: n_gpus', r['n_gpus'], 'world_seen', r['world_size_seen'], 'value', round(r['value'],1))" | tee -a $O/threads.txt
