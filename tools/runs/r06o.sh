#!/bin/bash
# round 6, GPU call O: weight-gradient partial sums carried through a cascade's time-steps (ops.WgradParts) against one reduction per time-step; 16-row k_uconv_h tests
O=gpurun_out/r06o; mkdir -p $O
python3 -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_backward.py tests/test_gpu_unet_fused.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
: > $O/ab.txt
for v in 1 0 1 0; do
  MRIDC_AMD_TL_WGRAD_DEFER=$v python3 bench.py --train --dtype bf16 --no-cpu-baseline --no-other-configs --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('defer=$v', round(r['value'],2), r['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
