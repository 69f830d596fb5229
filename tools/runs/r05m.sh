#!/bin/bash
# round 5, GPU call M: k_uconv_h as resident workgroups walking their items (lib_v_uh_persist) against the product form: E2EVN line, alternating
O=gpurun_out/r05m; mkdir -p $O
for rep in 1 2 3; do
  for v in product persist; do
    if [ $v = persist ]; then export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uh_persist/libmridc_amd.so; else unset MRIDC_AMD_LIB; fi
    timeout 300 python bench.py --model e2evn --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', 'e2evn', round(r['value'],1), 'slices/s', round(r['ms_per_step'],3), 'ms', 'parity', (r.get('parity_vs_oracle') or {}).get('rel_l2'))" | tee -a $O/ab.txt
  done
done
export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uh_persist/libmridc_amd.so
timeout 600 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py -m gpu -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
