#!/bin/bash
# round 5, GPU call AC: the headline with the tap products pre-summed along x (MRIDC_AMD_RIM_TAPS_Q=1, default) against the 18-plane form, alternating on one box
O=gpurun_out/r05ac; mkdir -p $O
for rep in 1 2 3; do
  for v in 0 1; do
    MRIDC_AMD_RIM_TAPS_Q=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-stream-inputs 2> $O/err_$v.txt | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); b = r['breakdown_ms']; print('taps_q=$v', round(r['value'], 2), r['parity_vs_oracle']['rel_l2'] if r.get('parity_vs_oracle') else None, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in b.items() if k in ('llg', 'conv_layer1', 'conv_layer2', 'final')})" | tee -a $O/headline.txt
  done
done
