#!/bin/bash
# round 5, GPU call J: k_uconv_h with the XCD band map (product lib) against the lib without it (lib_v_uh_noband): E2EVN and qCIRIM lines, alternating
O=gpurun_out/r05j; mkdir -p $O
for rep in 1 2 3; do
  for v in band noband; do
    if [ $v = noband ]; then export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uh_noband/libmridc_amd.so; else unset MRIDC_AMD_LIB; fi
    for m in e2evn qcirim; do
      timeout 300 python bench.py --model $m --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', '$m', round(r['value'],1), 'slices/s', round(r['ms_per_step'],3), 'ms', 'roofline', r.get('roofline',{}).get('avg_ms'))" | tee -a $O/ab.txt
    done
  done
done
unset MRIDC_AMD_LIB
timeout 600 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py -m gpu -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
