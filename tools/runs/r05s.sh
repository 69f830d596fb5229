#!/bin/bash
# round 5, GPU call S: k_uconvT with its plane loads one round ahead (product) against the plain loop (lib_v_uct_old): E2EVN line + U-Net tests
O=gpurun_out/r05s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py -m gpu -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uct_old/libmridc_amd.so; else unset MRIDC_AMD_LIB; fi
    timeout 300 python bench.py --model e2evn --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', 'e2evn', round(r['value'],1), 'slices/s', round(r['ms_per_step'],3), 'ms')" | tee -a $O/ab.txt
  done
done
