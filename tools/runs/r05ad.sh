#!/bin/bash
# round 5, GPU call AD: k_uconvT with its weights through the scalar cache (SGPR operands) against the LDS broadcast reads -- U-Net tests, then the E2EVN line alternating
O=gpurun_out/r05ad; mkdir -p $O
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uct_sgpr/libmridc_amd.so timeout 600 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest.txt
for rep in 1 2; do
  for v in uct_lds uct_sgpr; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 600 python bench.py --model e2evn --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$v', round(r['value'], 1), r['ms_per_step'])" | tee -a $O/e2evn.txt
  done
done
