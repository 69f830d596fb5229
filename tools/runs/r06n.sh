#!/bin/bash
# round 6, GPU call N: 16-row work items in k_uconv_h (lib) against the 8-row form (lib_v_th8): NormUnet tests, phase probe, E2EVN throughput
O=gpurun_out/r06n; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
: > $O/ab.txt
for v in lib lib_v_th8 lib lib_v_th8; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --model e2evn --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['value'],1), r.get('roofline'), r.get('breakdown_ms'))" >> $O/ab.txt
done
cat $O/ab.txt
