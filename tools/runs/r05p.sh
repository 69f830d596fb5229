#!/bin/bash
# round 5, GPU call P: the tap gather folded into the general-mask gradient -- tests, 2-D-mask line with the fold on / off
O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_graph.py tests/test_gpu_concurrent_streams.py -m gpu -q -x -k "general_mask or gather or graph or concurrent or mask" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-250
for rep in 1 2 3; do
  for f in 1 0; do
    MRIDC_AMD_LLG_T4_GATHER=$f timeout 300 python bench.py --mask 2d --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fold', $f, 'mask2d', round(r['value'],1), 'slices/s', round(r['ms_per_step'],2), 'ms', r.get('concurrent_replays_bit_identical_to_serial'))" | tee -a $O/ab.txt
  done
done
